# round 5, GPU batch 5: wide kernel with the K-block-pair order; per-level NMS; dense 1x1 through the wide kernel (A/B)
python -m pytest tests/test_conv_gpu.py tests/test_rcnn_ops_gpu.py -x -q -m gpu -k "igemm_wide or fwd_dgrad_wgrad or last_kernel or rpn_proposals or nms" 2>&1 | tail -15 > gpurun_out/r05_t5.log
( for knob in 3 32771; do echo "== bd_conv_set_patch3x3($knob)  [32771 = bit 15: generic kernel]"; BD_KNOB=$knob python scripts/micro_s2.py fwd dgrad 2>&1 | grep -v amdgpu | tail -8; done ) > gpurun_out/r05_s2_micro.txt
( echo "# d1 = default dense 1x1 dispatch; d0 under bd_conv_set_patch3x3(65536) = every dense 1x1 launch on conv_igemm_wide_kernel"; BD_KNOB=65536 python scripts/micro_1x1_step.py 1 0 2>&1 | grep -v amdgpu ) > gpurun_out/r05_dense1x1_wide.txt
B="python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --ref-protocol-steps 0"
for rep in 1 2; do
  for k in 3 32771; do
    $B --conv-knob $k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('retinanet_r50 conv-knob $k', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05_wide_ab.txt
$B --workload faster_rcnn_r50_800x1344 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('faster_rcnn per-level NMS', d['value'], d['ms_per_step'])" > gpurun_out/r05_frcnn_nms.txt
python -m pytest tests/test_groupnorm_gpu.py tests/test_fullsize_gpu.py::test_fcos_training_step_is_bitwise_reproducible_full_size -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/r05_t5.log
$B --workload fcos_r50_800x1344 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fcos (parallel GN finals)', d['value'], d['ms_per_step'])" > gpurun_out/r05_fcos_gn2.txt
