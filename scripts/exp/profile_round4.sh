# ROUND-4 collection script, kept for the record of how profiles/r04_* were made.  Its BD_DENSE1X1_* environment prefixes act only on a
# -DBD_TUNING build of the library (BD_EXTRA_FLAGS=-DBD_TUNING python -m basedet_amd.build): on the shipped library they are ignored and those
# lines measure the default twice.  The current script is scripts/profile_round6.sh.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p
mkdir -p $O
rm -f $O/workloads.txt
cd $R
B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --ref-protocol-steps 0 --serial-wgrad"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/bench_under_rocprof.json 2> $O/kt.err
P="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-pmc --ref-protocol-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $P > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $P > /dev/null 2> $O/write.err
python3 scripts/pmc_traffic.py $O/fetch $O/write $O/r04_pmc_traffic.json > $O/traffic.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $P --serial-wgrad > /dev/null 2> $O/sq.err
python3 scripts/pmc_sq.py $O/sq $O/r04_pmc_sq.json > $O/sq.txt 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_bench_kernel_stats.csv
rm -rf $O/kt $O/fetch $O/write $O/sq
python3 bench.py > $O/r04_bench.json 2> $O/bench.err
for w in fcos_r50_800x1344 faster_rcnn_r50_800x1344 atss_r50_800x1344 ota_r50_800x1344 freeanchor_r50_800x1344 retinanet_r101_800x1344; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])" >> $O/workloads.txt
done
python3 bench.py --workload retinanet_r101_800x1344 --fp8 --steps 20 --warmup 5 --no-cpu-baseline --ref-protocol-steps 0 > $O/r04_bench_r101_fp8.json 2>/dev/null
python3 bench.py --workload retinanet_r101_800x1344 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --ref-protocol-steps 0 > $O/r04_bench_r101_bf16_b32.json 2>/dev/null
python3 bench.py --workload fcos_r50_800x1344 --steps 50 --warmup 10 --no-cpu-baseline --ref-protocol-steps 0 > $O/r04_bench_fcos_r50.json 2>/dev/null
python3 bench.py --workload faster_rcnn_r50_800x1344 --steps 50 --warmup 10 --no-cpu-baseline --ref-protocol-steps 0 > $O/r04_bench_faster_rcnn_r50.json 2>/dev/null
for a in "--batch 16" "--batch 32" "--batch 16 --fp8" "--batch 32 --fp8" "--batch 32 --fp8 --model-opt FP8_DGRAD=0" "--batch 16 --model-opt FUSE_FROZEN_BLOCKS=0" "--batch 16 --conv-knob 24579" "--batch 16 --wgrad-knob 5" "--batch 16 --model-opt WGRAD_QUEUE=layer" "--batch 16"; do
  python3 bench.py $a --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet_r50 $a', d['value'], d['ms_per_step'])" >> $O/workloads.txt
done
BD_DENSE1X1_RING=0 python3 bench.py --batch 16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet_r50 --batch 16 BD_DENSE1X1_RING=0', d['value'], d['ms_per_step'])" >> $O/workloads.txt
python3 bench.py --batch 16 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet_r50 --batch 16 (again)', d['value'], d['ms_per_step'])" >> $O/workloads.txt
for a in "--batch 32 --fp8 --model-opt FP8_DGRAD=0" "--batch 16" "--batch 16 --wgrad-knob 5"; do
  python3 bench.py --workload retinanet_r101_800x1344 $a --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('retinanet_r101 $a', d['value'], d['ms_per_step'])" >> $O/workloads.txt
done
BD_FORCE_ALLREDUCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --ref-protocol-steps 0 2>$O/torchrun.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun world1 forced allreduce', d['value'], d['ms_per_step'])" >> $O/workloads.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/exp/ingest_rate_dma.hip -o /tmp/ingest_rate_dma 2>/dev/null && /tmp/ingest_rate_dma > $O/r04_ingest_rate.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/exp/ingest_rate_reg.hip -o /tmp/ingest_rate_reg 2>/dev/null && /tmp/ingest_rate_reg >> $O/r04_ingest_rate.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/exp/ingest_rate_mix.hip -o /tmp/ingest_rate_mix 2>/dev/null && /tmp/ingest_rate_mix >> $O/r04_ingest_rate.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/exp/epi_rows.hip -o /tmp/epi_rows 2>/dev/null && /tmp/epi_rows >> $O/r04_ingest_rate.txt
( echo "== default"; python3 scripts/micro_1x1_step.py 1 4 2 2>&1 | grep -v amdgpu; for d in 2 3 4; do echo "== no LDS-DMA variant, register sets in flight: $d"; BD_DENSE1X1_DMA_K=99999 BD_DENSE1X1_REGDEPTH=$d python3 scripts/micro_1x1_step.py 1 2>&1 | grep -v amdgpu; done ) > $O/r04_dense1x1_variants.txt
( echo "# scripts/micro_1x1_step.py 1 5 3: d1 = the default dispatch, d5 = conv1x1_ring_kernel for every legal launch, d3 = conv1x1_dense_kernel only; b = bit-packed gates"; python3 scripts/micro_1x1_step.py 1 5 3 2>&1 | grep -v amdgpu; echo "# phase stamps of conv1x1_ring_kernel (scripts/exp/r1x_stamp.py, -DBD_R1X_STAMP build)"; BASEDET_HIP_LIB=$R/basedet_amd/lib/libbasedet_r1x.so python3 scripts/exp/r1x_stamp.py 2>&1 | grep -v amdgpu ) > $O/r04_dense1x1_ring_final.txt
python3 scripts/micro_wgrad1x1_ring.py 5 2>&1 | grep -v amdgpu > $O/r04_wgrad1x1_ring_vs_staged.txt
python3 scripts/micro_wgrad_ring.py 5 2>&1 | grep -v amdgpu > $O/r04_wgrad3x3_ring_vs_staged.txt
BASEDET_HIP_LIB=$R/basedet_amd/lib/libbasedet_rk.so python3 scripts/exp/rk_stamp.py 2>&1 | grep -v amdgpu > $O/r04_rk_stamps.txt
cat $O/workloads.txt; cat $O/traffic.txt | head -12; cat $O/sq.txt | head; head -c 600 $O/r04_bench.json
