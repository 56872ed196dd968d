# round 5, GPU batch 6: Faster R-CNN with the RPN head's backward on the side stream
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "igemm_wide" 2>&1 | tail -3 > gpurun_out/r05_t6.log
python -m pytest tests/test_wgrad_queue_gpu.py tests/test_fullsize_parity_gpu.py::test_faster_rcnn_r50_full_size_matches_oracle tests/test_bench_batch_gpu.py::test_faster_rcnn_bench_batch_equals_tiled_batch2 tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/r05_t6.log
B="python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --ref-protocol-steps 0"
for rep in 1 2; do
  for opt in "RPN_BWD_SIDE=1" "RPN_BWD_SIDE=0"; do
    $B --workload faster_rcnn_r50_800x1344 --model-opt $opt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('faster_rcnn $opt', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05_frcnn_ab2.txt
