cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tr; mkdir -p $O; cd $R
for w in ${WL:-retinanet_r50_800x1344 fcos_r50_800x1344}; do
rocprofv3 --kernel-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0 > /dev/null 2> $O/kt_$w.err
find $O/kt_$w -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/${w}_trace.csv
rm -rf $O/kt_$w
done
