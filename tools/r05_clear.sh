#!/bin/bash
# C3 fit with the early clear of the panels done by the runtime's memset (0) or by N resident workgroups: tools/r05_clear.sh
mkdir -p gpurun_out/r05
for W in ${WGS:-0 32 64 128}; do
  SPLPAK_ND_CLEAR_WGS=$W timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-legs --neval 1000000 > gpurun_out/r05/clear_$W.json 2> gpurun_out/r05/clear_$W.err || exit 1
  python - "$W" <<'PY'
import json, sys
w = sys.argv[1]
d = json.loads(open(f"gpurun_out/r05/clear_{w}.json").read().strip().splitlines()[-1])
a = d.get("assembly", {})
print(f"clear wgs {w:>4}: {d['ms_per_step']:.2f} ms/fit, phases {d['config']['phase_seconds_per_step']}, "
      + ", ".join(f"{k.split(' (')[0]} {v['ms']:.2f}" for k, v in a.items() if isinstance(v, dict) and 'ms' in v))
PY
done
