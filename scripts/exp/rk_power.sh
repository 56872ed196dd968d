#!/bin/bash
# conv_wgrad3x3_ring_kernel: what its operand traffic costs (would a smaller input halo / a shorter load segment pay?).
# Diagnostic builds -DBD_RK_ABLATE=<bits> (TIMING ONLY): 1 = half the transposing reads, 2 = no operand DMA in the loop, 4 = X pieces 0-7 of 13 only.
# Build:  for a in 0 1 2 4; do BD_LIB_NAME=libbd_rk_abl$a.so BD_EXTRA_FLAGS="-DBD_RK_ABLATE=$a" python -m basedet_amd.build; done
# Run:    bash scripts/exp/rk_power.sh > gpurun_out/rk_power.txt
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for a in 0 1 2 4; do
    echo "== rep $rep BD_RK_ABLATE=$a"
    BASEDET_HIP_LIB=$PWD/basedet_amd/lib/libbd_rk_abl$a.so python3 scripts/micro_wgrad_ring.py 3 2>&1 | grep -E "head tower|res4 conv2|fpn out 256->256 100x168"
  done
done
