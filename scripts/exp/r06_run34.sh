timeout 900 python3 -m pytest tests/test_groupnorm_gpu.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python3 -m pytest tests/test_model_gpu.py tests/test_fullsize_parity_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "fcos or atss or ota" 2>&1 | tail -5
bash scripts/exp/ab_knob.sh "--workload fcos_r50_800x1344" "--workload fcos_r50_800x1344 --model-opt FUSE_GN_STATS=0" 2>&1 | tee gpurun_out/r06_gn_fused_ab.txt
