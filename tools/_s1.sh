timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_conv.py -q -x > gpurun_out/r5f_t_fp8.log 2>&1 || { tail -30 gpurun_out/r5f_t_fp8.log; exit 1; }
tail -1 gpurun_out/r5f_t_fp8.log
BENCH_ARGS="--fpn --fp8 --batch-per-gpu 8" AB_LIST="FRCNN_LIB=lib2dod_hip.so FRCNN_LIB=lib2dod_hip_prev.so FRCNN_LIB=lib2dod_hip.so FRCNN_LIB=lib2dod_hip_prev.so" bash tools/ab_lib.sh > gpurun_out/r5f_ab13.txt 2>&1
cat gpurun_out/r5f_ab13.txt
