python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "thin" 2>&1 | tail -6
python -m pytest tests/test_model_gpu.py tests/test_wgrad_queue_gpu.py tests/test_fullsize_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu -k "rcnn" 2>&1 | tail -6
for r in 1 2; do
for f in "" "--model-opt THIN_RPN_BWD=0"; do
python bench.py --workload faster_rcnn_r50_800x1344 --steps 20 --warmup 5 --no-roofline $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"
done; done
