cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
for k in 3 16387 3 16387; do
python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 --conv-knob $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('knob $k', d['value'], d['ms_per_step'])"
done
BASEDET_HIP_LIB=basedet_amd/lib/libbd_ppstamp.so python3 scripts/exp/pp_clock.py 3 2>&1 | grep -v amdgpu.ids
