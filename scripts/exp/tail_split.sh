cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
for k in 3 8195 3 8195; do
python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 --conv-knob $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('knob $k', d['value'], d['ms_per_step'])"
done
for k in 3 8195 3 8195; do
python3 bench.py --workload retinanet_r101_800x1344 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 --conv-knob $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('r101 knob $k', d['value'], d['ms_per_step'])"
done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --ref-protocol-steps 0 --no-pmc --dump-convs --serial-wgrad > gpurun_out/ts_3.json 2> gpurun_out/ts_3.err
grep "R=3 s=1" gpurun_out/ts_3.err | grep "50x84"
