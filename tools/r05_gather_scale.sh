#!/bin/bash
# stencil_gather / gram_wave time against the grid size (does the gather run at the same rate when its blocks fit the 256 MB
# Infinity Cache?): tools/r05_gather_scale.sh
for n in 24 32 48 64; do
  out="$GRAFT_REPO_ROOT/gpurun_out/r05/gs_$n"; rm -rf "$out"; mkdir -p "$out"
  m=$((n*n*n*38))
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/tools/nd_repeat.py" 3 $n $m 3 > "$out.log" 2>&1)
  f=$(find "$out" -name "*kernel_stats.csv" | head -1)
  echo "== $n^3 nodes, $m points"; grep -E "stencil_gather|gram_wave|sp_scatter|nd_init" "$f" | cut -d, -f1-4 | cut -c1-150
done
