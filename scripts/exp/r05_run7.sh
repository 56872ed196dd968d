# round 5, GPU batch 7: which change broke test_faster_rcnn_bench_batch_equals_tiled_batch2?
for cfg in "1" "0"; do
  echo "== bd_rpn_set_nms_per_level($cfg), RPN_BWD_SIDE default (0)" >> gpurun_out/r05_t7.log
  BD_TEST_NMS_PER_LEVEL=$cfg python -m pytest tests/test_bench_batch_gpu.py::test_faster_rcnn_bench_batch_equals_tiled_batch2 -x -q -m gpu 2>&1 | tail -12 >> gpurun_out/r05_t7.log
done
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "dense_1x1 or last_kernel or igemm_wide" 2>&1 | tail -3 >> gpurun_out/r05_t7.log
