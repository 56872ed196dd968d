cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/frp; mkdir -p $O; cd $R
w=faster_rcnn_r50_800x1344
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0 --serial-wgrad > /dev/null 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/frcnn_kernel_stats.csv
rm -rf $O/kt
