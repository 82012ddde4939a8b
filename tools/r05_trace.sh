#!/bin/bash
# Launch timeline of the assembly of the last C3 fit under rocprofv3 --kernel-trace: tools/r05_trace.sh <tag> [lines]
# (environment switches are taken from the caller's environment)
tag=$1; lines=${2:-30}
out="$GRAFT_REPO_ROOT/gpurun_out/r05/trace_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/tools/nd_repeat.py" 3 64 10000000 3 > "$out.log" 2>&1
f=$(ls -S $(find "$out" -name "*kernel_trace.csv") | head -1)
cp "$f" "$out.kernel_trace.csv"
python3 "$GRAFT_REPO_ROOT/tools/last_fit_trace.py" "$f" > "$out.timeline.txt"
head -$lines "$out.timeline.txt"
