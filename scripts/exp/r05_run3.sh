# round 5, GPU batch 3: Faster R-CNN tests + same-box A/B of the early RPN targets; FCOS kernel profile; C2 per-shape table
python -m pytest tests/test_wgrad_queue_gpu.py tests/test_rcnn_ops_gpu.py tests/test_fullsize_parity_gpu.py::test_faster_rcnn_r50_full_size_matches_oracle tests/test_bench_batch_gpu.py::test_faster_rcnn_bench_batch_equals_tiled_batch2 -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_t3.log
B="python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --ref-protocol-steps 0"
for rep in 1 2; do
  for opt in "RPN_TARGETS_EARLY=1" "RPN_TARGETS_EARLY=0"; do
    $B --workload faster_rcnn_r50_800x1344 --model-opt $opt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('faster_rcnn $opt', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05_frcnn_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fcos -o fcos -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcos_r50_800x1344 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --ref-protocol-steps 0 --serial-wgrad > $GRAFT_REPO_ROOT/gpurun_out/r05_fcos_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_fcos -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_fcos_kernel_stats.csv \;
find gpurun_out/prof_fcos -name "*kernel_trace.csv" -delete
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ref-protocol-steps 0 --no-pmc --dump-convs > gpurun_out/r05_bench_dump.json 2> gpurun_out/r05_bench_dump.err
