#!/bin/bash
# Round-3 profile set on the GPU box (results under gpurun_out/r03/): bench line, rocprofv3 kernel stats of the same
# command, PMC traffic passes (FETCH_SIZE, WRITE_SIZE separately) of one C3 fit and of the evaluation kernels.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
python bench.py --steps 5 --warmup 1 > gpurun_out/r03/c3_bench.json 2> gpurun_out/r03/c3_bench.err || exit 1
echo "bench done"
bash tools/prof.sh r03/bench_stats python3 /root/repo/bench.py --steps 5 --warmup 1 --no-side-legs --no-cpu-baseline > /dev/null
echo "stats done"
export C2_WARM=0 C2_REPS=1
bash tools/pmc.sh r03/fit_fetch "FETCH_SIZE" "nd_|gram|gather|residual|scatter" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
bash tools/pmc.sh r03/fit_write "WRITE_SIZE" "nd_|gram|gather|residual|scatter" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
echo "fit pmc done"
bash tools/pmc.sh r03/eval3_fetch "FETCH_SIZE" "eval|bin_|run_place" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r03/eval3_write "WRITE_SIZE" "eval|bin_|run_place" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r03/eval4_fetch "FETCH_SIZE" "eval|bin_|run_place" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
bash tools/pmc.sh r03/eval4_write "WRITE_SIZE" "eval|bin_|run_place" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
echo "eval pmc done"
ls gpurun_out/r03
python3 tools/r03_pmc_json.py gpurun_out/r03 gpurun_out/r03 > gpurun_out/r03/pmc_json.log 2>&1
