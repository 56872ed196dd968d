#!/bin/bash
# Same-box A/B of several builds of the library over the default bench (alternating, three repetitions):
#   bash scripts/exp/ab_libs.sh <libA.so> <libB.so> [...] [-- bench args...]
cd "$(dirname "$0")/../.."
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2 3; do
  for L in "${LIBS[@]}"; do
    BASEDET_HIP_LIB=$PWD/basedet_amd/lib/$L python3 bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --ref-protocol-steps 0 "$@" 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L rep $rep', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
