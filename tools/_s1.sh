python -c "import torch; print(torch.cuda.Stream.priority_range())"
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "nms" 2>&1 | tail -1
AB_LIST="FRCNN_LIB=lib2dod_hip.so FRCNN_LIB=lib2dod_hip.so+FRCNN_SIDE_PRIORITY=1 FRCNN_LIB=lib2dod_hip.so+FRCNN_SIDE_PRIORITY=0 FRCNN_LIB=lib2dod_hip.so FRCNN_LIB=lib2dod_hip.so+FRCNN_SIDE_PRIORITY=1" bash tools/ab_lib.sh > gpurun_out/r5e_ab9.txt 2>&1
cat gpurun_out/r5e_ab9.txt
