# round 5, GPU batch 4: conv_igemm_wide_kernel -- parity, per-launch timing against the generic kernel, step A/B; FCOS kernel profile
python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "igemm_wide or fwd_dgrad_wgrad or last_kernel or strided_dgrad" 2>&1 | tail -15 > gpurun_out/r05_t4.log
( for knob in 3 32771; do echo "== bd_conv_set_patch3x3($knob)  [32771 = bit 15: generic kernel]"; BD_KNOB=$knob python scripts/micro_s2.py fwd dgrad 2>&1 | grep -v amdgpu; done ) > gpurun_out/r05_s2_micro.txt
B="python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --ref-protocol-steps 0"
for rep in 1 2; do
  for k in 3 32771; do
    $B --conv-knob $k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('retinanet_r50 conv-knob $k', d['value'], d['ms_per_step'])"
  done
done > gpurun_out/r05_wide_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fcos -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcos_r50_800x1344 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --ref-protocol-steps 0 --serial-wgrad > $GRAFT_REPO_ROOT/gpurun_out/r05_fcos_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_fcos -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_fcos_kernel_stats.csv
rm -rf gpurun_out/prof_fcos
