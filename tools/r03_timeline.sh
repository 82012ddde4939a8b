#!/bin/bash
# Kernel trace of one C3 fit -> per-depth timeline of the nested-dissection factorisation, and the refinement log.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
export C2_WARM=1 C2_REPS=1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r03/trace" -- python3 /root/repo/tools/c2_profile.py 3 64 10000000 > "$GRAFT_REPO_ROOT/gpurun_out/r03/trace.log" 2>&1 )
f=$(find gpurun_out/r03/trace -name "*kernel_trace.csv" | head -1)
python3 tools/nd_timeline.py "$f" > gpurun_out/r03/timeline.txt 2>&1
rm -rf gpurun_out/r03/trace
tail -25 gpurun_out/r03/timeline.txt
SPLPAK_DEBUG=1 python3 tools/c2_profile.py 3 64 10000000 2>&1 | grep -i "refinement step" | tail -4
