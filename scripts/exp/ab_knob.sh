#!/bin/bash
# Same-box A/B of bench.py argument sets (alternating, three repetitions):  bash scripts/exp/ab_knob.sh "<args A>" "<args B>" [...]
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for A in "$@"; do
    python3 bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --ref-protocol-steps 0 $A 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$A] rep $rep', d['value'], 'img/s', d['ms_per_step'], 'ms')"
  done
done
