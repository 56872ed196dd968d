# Round 6: the stability protocol of scripts/exp/fp8_stability.sh (R101, batch 32, ONE repeated batch, 1 500 steps after 20 warm-up, five seeds) at
# --lr-scale 1.0 -- the configured learning rate, where round 3 found bf16 itself at its stability edge -- with the rebuilt conv3x3_pp8_kernel:
# bf16 against the --fp8 default (e4m3 forward, e5m2 data gradients, one-byte 3x3 weight gradients, per-group delayed scales).
# usage: bash scripts/exp/fp8_stability_r6.sh [steps=1500] [lr_scale=1.0] [seeds="0 1 2 3 4"]
STEPS=${1:-1500}; LR=${2:-1.0}; SEEDS=${3:-"0 1 2 3 4"}
O=gpurun_out/r06_fp8_stability_lr$LR.txt
echo "# R101 batch 32, $STEPS repeated-batch steps after 20 warm-up steps, --lr-scale $LR; loss every 250 steps, final loss, img/s" > $O
for seed in $SEEDS; do
  for mode in "bf16|" "fp8 default (e5m2 dgrad + one-byte 3x3 wgrad, group scales)|--fp8"; do
    name=${mode%%|*}; flags=${mode#*|}
    python bench.py --workload retinanet_r101_800x1344 --batch 32 $flags --steps $STEPS --warmup 20 --seed $seed --lr-scale $LR --log-every 250 \
        --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0 > /tmp/st.json 2> /tmp/st.err
    rc=$?
    losses=$(grep "^# step" /tmp/st.err | awk '{printf "%s ", $5}')
    val=$(python -c "import json; d=json.load(open('/tmp/st.json')); print(d['config'].get('final_loss'), d['value'])" 2>/dev/null)
    echo "seed $seed | $name | rc $rc | losses $losses| final/img_s $val" >> $O
    tail -1 $O
  done
done
