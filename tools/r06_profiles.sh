#!/bin/bash
# Round-6 profile set on the GPU box (results under gpurun_out/r06/): bench line, rocprofv3 kernel stats of the same
# command, PMC traffic passes (FETCH_SIZE, WRITE_SIZE separately) of one C3 fit and of the evaluation kernels.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
python bench.py --steps 10 --warmup 2 > gpurun_out/r06/c3_bench.json 2> gpurun_out/r06/c3_bench.err || exit 1
echo "bench done"
bash tools/prof.sh r06/bench_stats python3 /root/repo/bench.py --steps 5 --warmup 1 --no-side-legs --no-cpu-baseline > /dev/null
echo "stats done"
export C2_WARM=0 C2_REPS=1
bash tools/pmc.sh r06/fit_fetch "FETCH_SIZE" "nd_|gram|gather|residual|scatter" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
bash tools/pmc.sh r06/fit_write "WRITE_SIZE" "nd_|gram|gather|residual|scatter" python3 /root/repo/tools/c2_profile.py 3 64 10000000 > /dev/null
echo "fit pmc done"
bash tools/pmc.sh r06/eval3_fetch "FETCH_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r06/eval3_write "WRITE_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r06/eval3_valu "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
bash tools/pmc.sh r06/eval4_fetch "FETCH_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
bash tools/pmc.sh r06/eval4_write "WRITE_SIZE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
bash tools/pmc.sh r06/eval4_valu "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "eval|bin_|run_place|pr_" python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
EVAL_PROFILE_REPS=20 bash tools/prof.sh r06/eval3_stats python3 /root/repo/tools/eval_profile.py 3 64 50000000 > /dev/null
EVAL_PROFILE_REPS=6 bash tools/prof.sh r06/eval4_stats python3 /root/repo/tools/eval_profile.py 4 32 100000000 > /dev/null
echo "eval pmc done"
export C2_WARM=3 C2_REPS=10
bash tools/prof.sh r06/c2_stats python3 /root/repo/tools/c2_profile.py 2 64 1000000 > /dev/null
echo "c2 stats done"
ls gpurun_out/r06
# config 5's fit at its own size by the iterative solve (round 6): kernel stats and the traffic of its kernels (one fit each)
export C2_WARM=0 C2_REPS=1
bash tools/prof.sh r06/c5_pcg_stats python3 /root/repo/tools/c2_profile.py 4 32 10000000 > /dev/null
bash tools/pmc.sh r06/c5_pcg_fetch "FETCH_SIZE" "rows4|tri_pass|mode_p|bj_apply|dot_|update_" python3 /root/repo/tools/c2_profile.py 4 32 10000000 > /dev/null
bash tools/pmc.sh r06/c5_pcg_write "WRITE_SIZE" "rows4|tri_pass|mode_p|bj_apply|dot_|update_" python3 /root/repo/tools/c2_profile.py 4 32 10000000 > /dev/null
bash tools/pmc.sh r06/c5_pcg_valu "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS" "rows4_tile" python3 /root/repo/tools/c2_profile.py 4 32 10000000 > /dev/null
echo "c5 pcg done"
ls gpurun_out/r06
