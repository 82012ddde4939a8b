# kernel trace of C3 fits with a given SPLPAK_ND_HALVES; prints the holes of the last factorisation.   tools/nd/trace_halves.sh <halves>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
export C2_WARM=1 C2_REPS=1 SPLPAK_ND_HALVES=$1 SPLPAK_SOLVER=direct GPU_MAX_HW_QUEUES=2
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r6/trace_h$1" -- python3 /root/repo/tools/c2_profile.py 3 64 10000000 > "$GRAFT_REPO_ROOT/gpurun_out/r6/trace_h$1.log" 2>&1 )
f=$(find gpurun_out/r6/trace_h$1 -name "*kernel_trace.csv" | head -1)
python3 tools/nd/holes.py "$f" 200 > gpurun_out/r6/holes_h$1.txt 2>&1
python3 tools/nd_timeline.py "$f" >> gpurun_out/r6/holes_h$1.txt 2>&1
rm -rf gpurun_out/r6/trace_h$1
