#!/bin/bash
# kernel trace of a few fits (default: BASELINE config 2) -> launch list of the last one
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
export C2_WARM=3 C2_REPS=3
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r05/c2trace" -- python3 /root/repo/tools/c2_profile.py ${1:-2} ${2:-64} ${3:-1000000} > "$GRAFT_REPO_ROOT/gpurun_out/r05/c2trace.log" 2>&1 )
f=$(find gpurun_out/r05/c2trace -name "*kernel_trace.csv" | head -1)
python3 tools/last_fit_trace.py "$f" > gpurun_out/r05/c2_last_fit.txt 2>&1
rm -rf gpurun_out/r05/c2trace
tail -3 gpurun_out/r05/c2_last_fit.txt
