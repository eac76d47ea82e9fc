set -o pipefail
PMC_TABLE=1 timeout -k 10 400 bash tools/profile_step.sh r05_g > gpurun_out/r5r_prof1.log 2>&1; echo "prof1 rc=$?"
BENCH_ARGS="--depth 101 --proposals 1000 --batch-per-gpu 2" OFFLINE_VARIANT=_r101 timeout -k 10 300 bash tools/profile_step.sh r05_g_r101 > gpurun_out/r5r_prof2.log 2>&1; echo "prof2 rc=$?"
BENCH_ARGS="--fpn --fp8 --batch-per-gpu 8" OFFLINE_VARIANT=_fp8_fpn timeout -k 10 300 bash tools/profile_step.sh r05_g_fp8_fpn > gpurun_out/r5r_prof3.log 2>&1; echo "prof3 rc=$?"
BENCH_ARGS="--image-shape 600 1987 --batch-per-gpu 2" OFFLINE_VARIANT=_ref600 timeout -k 10 300 bash tools/profile_step.sh r05_g_ref600 > gpurun_out/r5r_prof4.log 2>&1; echo "prof4 rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r5r_t_all.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5r_t_all.log
