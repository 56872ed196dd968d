# third matrix (see fp8_stability.sh / fp8_stability2.sh): the fp8 default with the 3x3 weight gradients from the one-byte twins as well (FP8_WGRAD=2)
STEPS=${1:-1500}; LRS=${2:-"0.25 0.5"}; SEEDS=${3:-"0 1 2 3 4"}
O=gpurun_out/r04_fp8_stability_wgrad.txt
echo "# R101 batch 32, $STEPS repeated-batch steps after 20 warm-up steps; --fp8 --model-opt FP8_WGRAD=2; loss every 250 steps, final loss, img/s" > $O
for lr in $LRS; do
for seed in $SEEDS; do
    python bench.py --workload retinanet_r101_800x1344 --batch 32 --fp8 --model-opt FP8_WGRAD=2 --steps $STEPS --warmup 20 --seed $seed --lr-scale $lr --log-every 250 \
        --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0 > /tmp/st.json 2> /tmp/st.err
    rc=$?
    losses=$(grep "^# step" /tmp/st.err | awk '{printf "%s ", $5}')
    val=$(python -c "import json; d=json.load(open('/tmp/st.json')); print(d['config'].get('final_loss'), d['value'])" 2>/dev/null)
    echo "lr-scale $lr | seed $seed | fp8 + FP8_WGRAD=2 | rc $rc | losses $losses| final/img_s $val" >> $O
    tail -1 $O
done
done
