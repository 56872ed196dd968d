#!/bin/bash
# what bounds conv3x3_pp_kernel -- stalls or energy?  Three diagnostic builds of the library (-DBD_PP_STAMP; ablations
# are TIMING ONLY, results wrong) run the 16 x 22 400-pixel head tower launch back to back for ~3 s each, alternating, on ONE box, and print
# launch time AND the in-kernel clock (d s_memtime / d s_memrealtime):
#   base  : the shipped loop with stamps
#   abl1  : half of the fragment ds_read_b128 removed (BD_PP_ABLATE=1)
#   abl2  : the weight DMA removed from the K loop (BD_PP_ABLATE=2)
#   abl3  : both
# Build (in the container, cross-compiled):  for a in 0 1 2 3; do BD_LIB_NAME=libbd_pp_abl$a.so BD_EXTRA_FLAGS="-DBD_PP_STAMP -DBD_PP_ABLATE=$a" python -m basedet_amd.build; done
# Run (GPU box): bash scripts/exp/pp_power.sh > gpurun_out/pp_power.txt
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for a in 0 1 2 3; do
    echo "== rep $rep BD_PP_ABLATE=$a"
    BASEDET_HIP_LIB=$PWD/basedet_amd/lib/libbd_pp_abl$a.so python3 scripts/exp/pp_clock.py 2>&1 | grep -E "head conv fwd|whole workgroup|tile 0:"
  done
done
