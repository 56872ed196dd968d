# Stability matrix of BASELINE config 5 (RetinaNet-R101, 32 images of 800x1344 per GPU) on ONE repeated batch: bf16 vs fp8 (e4m3 forward +
# e5m2 data gradients under per-group delayed scales: the --fp8 default) vs fp8 forward only, five seeds each, at a learning rate on which
# the bf16 reference itself is stable (round 3's protocol sat at bf16's own stability edge: one seed in three diverged in EVERY mode).
# usage: bash scripts/exp/fp8_stability.sh [steps=1500] [lr_scale=0.5] [seeds="0 1 2 3 4"]
STEPS=${1:-1500}; LR=${2:-0.5}; SEEDS=${3:-"0 1 2 3 4"}
O=gpurun_out/r04_fp8_stability.txt
echo "# R101 batch 32, $STEPS repeated-batch steps after 20 warm-up steps, --lr-scale $LR; loss every 250 steps, final loss, img/s" > $O
for seed in $SEEDS; do
  for mode in "bf16|" "fp8 default (e5m2 dgrad, group scales)|--fp8" "fp8 forward only|--fp8 --model-opt FP8_DGRAD=0"; do
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
