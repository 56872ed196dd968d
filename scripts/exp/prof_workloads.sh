# rocprofv3 kernel statistics of the secondary workloads (profiles/r03_<workload>_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for w in "faster_rcnn_r50_800x1344:faster_rcnn_r50:" "fcos_r50_800x1344:fcos_r50:" "retinanet_r101_800x1344:r101_fp8:--fp8 --batch 32"; do
  wl=${w%%:*}; rest=${w#*:}; name=${rest%%:*}; extra=${rest#*:}
  O=$R/gpurun_out/prof_$name
  mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --workload $wl $extra --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline --ref-protocol-steps 0 --serial-wgrad > $O/bench.json 2> $O/kt.err
  find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r03_${name}_kernel_stats.csv
  rm -rf $O/kt
  cut -c1-160 $O/bench.json
done
