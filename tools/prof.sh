#!/bin/bash
# Kernel statistics of a command on the GPU box: tools/prof.sh <outdir-under-gpurun_out> <program and args...>
# (rocprofv3 --kernel-trace --stats, CSV output; the program itself follows `--`, no wrapper in between)
out="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- "$@" > "$out.log" 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out.kernel_stats.csv" && head -${PROF_LINES:-25} "$f" | cut -c1-220
