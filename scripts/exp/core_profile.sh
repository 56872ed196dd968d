set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c
rm -rf $O; mkdir -p $O
cd $R
B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --ref-protocol-steps 0 --serial-wgrad"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/bench_under_rocprof.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_bench_kernel_stats.csv
rm -rf $O/kt
P="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $P --serial-wgrad > /dev/null 2> $O/sq.err
python3 scripts/pmc_sq.py $O/sq $O/r04_pmc_sq.json > $O/sq.txt 2>&1
rm -rf $O/sq
python3 bench.py > $O/r04_bench.json 2> $O/bench.err
head -c 400 $O/r04_bench.json; echo; cat $O/sq.txt | head -12; head -25 $O/r04_bench_kernel_stats.csv | cut -c1-150
