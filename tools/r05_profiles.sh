#!/bin/bash
# Round-5 profile set on the GPU box (results under gpurun_out/r05/): bench line, rocprofv3 kernel stats of the same
# command, PMC traffic passes (FETCH_SIZE, WRITE_SIZE separately) of one C3 fit and of the evaluation kernels.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05
python bench.py --steps 10 --warmup 2 > gpurun_out/r05/c3_bench.json 2> gpurun_out/r05/c3_bench.err || exit 1
echo "bench done"
bash tools/prof.sh r05/bench_stats python3 /root/repo/bench.py --steps 5 --warmup 1 --no-side-legs --no-cpu-baseline > /dev/null
echo "stats done"
export C2_WARM=0 C2_REPS=1
bash tools/pmc.sh r05/fit_fetch "FETCH_SIZE" "nd_|gram|gather|residual|scatter" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
bash tools/pmc.sh r05/fit_write "WRITE_SIZE" "nd_|gram|gather|residual|scatter" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
echo "fit pmc done"
bash tools/pmc.sh r05/eval3_fetch "FETCH_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r05/eval3_write "WRITE_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r05/eval3_valu "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r05/eval4_fetch "FETCH_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
bash tools/pmc.sh r05/eval4_write "WRITE_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
bash tools/pmc.sh r05/eval4_valu "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
EVAL_PROFILE_REPS=20 bash tools/prof.sh r05/eval3_stats python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
EVAL_PROFILE_REPS=6 bash tools/prof.sh r05/eval4_stats python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
echo "eval pmc done"
export C2_WARM=3 C2_REPS=10
bash tools/prof.sh r05/c2_stats python3 /root/repo/tools/c2_profile.py 2 64 1000000 > /dev/null
echo "c2 stats done"
ls gpurun_out/r05
