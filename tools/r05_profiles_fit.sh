#!/bin/bash
# The fit half of tools/r05_profiles.sh again (after the assembly work of round 5): bench line, kernel stats of the same
# command, PMC traffic passes of one C3 fit, kernel stats and launch list of config 2.  Results under gpurun_out/r05/.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
python bench.py --steps 10 --warmup 2 > gpurun_out/r05/c3_bench.json 2> gpurun_out/r05/c3_bench.err || exit 1
echo "bench done"
rm -rf gpurun_out/r05/bench_stats
bash tools/prof.sh r05/bench_stats python3 /root/repo/bench.py --steps 5 --warmup 1 --no-side-legs --no-cpu-baseline > /dev/null
echo "stats done"
export C2_WARM=0 C2_REPS=1
rm -rf gpurun_out/r05/fit_fetch gpurun_out/r05/fit_write
bash tools/pmc.sh r05/fit_fetch "FETCH_SIZE" "nd_|gram|gather|residual|scatter|sp_|constraint|backward" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
bash tools/pmc.sh r05/fit_write "WRITE_SIZE" "nd_|gram|gather|residual|scatter|sp_|constraint|backward" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
echo "fit pmc done"
export C2_WARM=3 C2_REPS=10
rm -rf gpurun_out/r05/c2_stats
bash tools/prof.sh r05/c2_stats python3 /root/repo/tools/c2_profile.py 2 64 1000000 > /dev/null
bash tools/c2_trace.sh > /dev/null
echo "c2 done"
ls gpurun_out/r05 | head -50
