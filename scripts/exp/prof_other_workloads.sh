cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4q
mkdir -p $O
cd $R
for w in fcos_r50_800x1344 faster_rcnn_r50_800x1344; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --ref-protocol-steps 0 --serial-wgrad > $O/${w}_under_rocprof.json 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_${w}_kernel_stats.csv
  rm -rf $O/kt_$w
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_r101 -- python3 bench.py --workload retinanet_r101_800x1344 --fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --ref-protocol-steps 0 --serial-wgrad > $O/r101_fp8_under_rocprof.json 2> $O/kt_r101.err
find $O/kt_r101 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_r101_fp8_kernel_stats.csv
rm -rf $O/kt_r101
ls -la $O
