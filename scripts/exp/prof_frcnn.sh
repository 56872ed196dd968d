cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/frcnn
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --workload faster_rcnn_r50_800x1344 $FRCNN_EXTRA --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-roofline --ref-protocol-steps 0 --serial-wgrad > $O/bench.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/frcnn_kernel_stats.csv
rm -rf $O/kt
cat $O/bench.json | cut -c1-200
