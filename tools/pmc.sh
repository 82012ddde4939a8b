#!/bin/bash
# Hardware counters of a command on the GPU box (own run, no tracing beside it):
#   tools/pmc.sh <outdir-under-gpurun_out> "<COUNTER COUNTER ...>" <kernel-regex> <program and args...>
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; ctr="$2"; rx="$3"; shift 3
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-include-regex "$rx" --output-format csv -d "$out" -- "$@" > "$out.log" 2>&1
f=$(find "$out" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out.counters.csv" && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" "$rx"
