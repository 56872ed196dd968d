cd $GRAFT_REPO_ROOT
for t in 512 768 1024 2048; do
BD_WGRAD3_TARGET=$t python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --ref-protocol-steps 0 --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('target $t', d['value'], d['ms_per_step'], [(k['kernel'],k['ms_per_step']) for k in d['roofline_others'] if 'wgrad3x3' in k['kernel']])"
done
