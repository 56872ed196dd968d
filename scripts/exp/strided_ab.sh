# VERDICT round 3, item 5 ("strided 3x3 off the generic kernel, or prove it does not matter"): same-box A/B of the step with the ten
# 3x3 / stride-2 forward + data-gradient launches (res3.0 / res4.0 / res5.0 conv2, P6, P7; conv_igemm_kernel) replaced by no-ops after the
# warm-up (bd_conv_desc.route[1] bit 4; outputs keep the last warm-up step's values).  The difference is the most a perfect kernel could win.
O=gpurun_out/r04_strided3x3_ab.txt
echo "# img/s and ms/step, alternating runs on one box: normal / strided 3x3 launches skipped (NOT a throughput claim)" > $O
for wl in "retinanet_r50_800x1344|16" "faster_rcnn_r50_800x1344|16" "retinanet_r101_800x1344|16"; do
  w=${wl%%|*}; b=${wl#*|}
  for rep in 1 2 3; do
    for skip in "" "--skip-s2-3x3-after-warmup"; do
      python bench.py --workload $w --batch $b --steps 40 --warmup 10 $skip --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'skipped' if '$skip' else 'normal ', d['value'], d['ms_per_step'], d['step_ms_p50'])" >> $O
    done
  done
done
cat $O
