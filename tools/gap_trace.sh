cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/gap
export C2_WARM=1 C2_REPS=1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/gap/trace" -- python3 /root/repo/tools/c2_profile.py 3 64 10000000 > "$GRAFT_REPO_ROOT/gpurun_out/gap/trace.log" 2>&1 )
f=$(find gpurun_out/gap/trace -name "*kernel_trace.csv" | head -1)
python3 tools/gap_report.py "$f" 30 > gpurun_out/gap/gaps.txt 2>&1
cp "$f" gpurun_out/gap/kernel_trace.csv
rm -rf gpurun_out/gap/trace
cat gpurun_out/gap/gaps.txt
