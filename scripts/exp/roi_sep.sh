cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rcnn_ops_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "rcnn or frcnn or faster" 2>&1 | tail -3
for v in 1 0 1 0; do
BD_ROI_BWD_SEP=$v python3 bench.py --workload faster_rcnn_r50_800x1344 --roi-bwd-pk --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pk sep $v', d['value'], d['ms_per_step'])"
done
