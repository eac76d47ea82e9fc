AB_LIST="FRCNN_LIB=lib2dod_hip.so FRCNN_LIB=lib2dod_hip.so+FRCNN_TARGETS_ON_RPN_SIDE=1 FRCNN_LIB=lib2dod_hip.so FRCNN_LIB=lib2dod_hip.so+FRCNN_TARGETS_ON_RPN_SIDE=1" bash tools/ab_lib.sh > gpurun_out/r5g_ab15.txt 2>&1
cat gpurun_out/r5g_ab15.txt
MODES="FRCNN_X=1 FRCNN_TARGETS_ON_RPN_SIDE=1 FRCNN_MAIN_FIRST=1" bash tools/prof_mode.sh r5g_fork > gpurun_out/r5g_fork_prof.txt 2>&1
for i in 1 2 3; do echo "== mode $i"; python tools/step_timeline.py gpurun_out/r5g_fork_${i}_kernel_trace.csv 20 | sed -n '/rpn_head_post/,/bn_bwd_apply/p' | cut -c1-110 | head -24; done
