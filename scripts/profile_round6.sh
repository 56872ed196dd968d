# Round-6 profile collection (one gpurun call): everything DESIGN.md section 5 quotes.  Outputs under gpurun_out/r6p/, copied to profiles/r06_*.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p
mkdir -p $O
rm -f $O/workloads.txt
cd $R
B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --ref-protocol-steps 0 --serial-wgrad"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/r06_bench_under_rocprof.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_bench_kernel_stats.csv
rm -rf $O/kt
P="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $P > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $P > /dev/null 2> $O/write.err
python3 scripts/pmc_traffic.py $O/fetch $O/write $O/r06_pmc_traffic.json > $O/traffic.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $P --serial-wgrad > /dev/null 2> $O/sq.err
python3 scripts/pmc_sq.py $O/sq $O/r06_pmc_sq.json > $O/sq.txt 2>&1
rm -rf $O/fetch $O/write $O/sq
python3 bench.py > $O/r06_bench.json 2> $O/bench.err
for w in fcos_r50_800x1344 faster_rcnn_r50_800x1344; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0 --serial-wgrad > /dev/null 2> $O/kt_$w.err
  find $O/kt_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_${w}_kernel_stats.csv
  rm -rf $O/kt_$w
  python3 bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --ref-protocol-steps 0 > $O/r06_bench_$w.json 2>/dev/null
done
for w in fcos_r50_800x1344 faster_rcnn_r50_800x1344 atss_r50_800x1344 ota_r50_800x1344 freeanchor_r50_800x1344 retinanet_r101_800x1344; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])" >> $O/workloads.txt
done
for a in "" ""; do
  python3 bench.py --workload faster_rcnn_r50_800x1344 $a --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('faster_rcnn_r50 $a', d['value'], d['ms_per_step'])" >> $O/workloads.txt
done
python3 bench.py --workload retinanet_r101_800x1344 --fp8 --steps 20 --warmup 5 --no-cpu-baseline --ref-protocol-steps 0 > $O/r06_bench_r101_fp8.json 2>/dev/null
python3 bench.py --workload retinanet_r101_800x1344 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --ref-protocol-steps 0 > $O/r06_bench_r101_bf16_b32.json 2>/dev/null
for a in "--batch 16" "--batch 32" "--batch 16 --fp8" "--batch 16 --model-opt WGRAD_QUEUE=layer" "--batch 16 --wgrad-knob 5" "--batch 16 --dense1x1 3" "--batch 16 --conv-knob 259" "--batch 16"; do
  python3 bench.py $a --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet_r50 $a', d['value'], d['ms_per_step'])" >> $O/workloads.txt
done
BD_FORCE_ALLREDUCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>$O/torchrun.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun world1 forced allreduce', d['value'], d['ms_per_step'])" >> $O/workloads.txt
cat $O/workloads.txt; cat $O/traffic.txt | head -12; cat $O/sq.txt | head; head -c 600 $O/r06_bench.json
