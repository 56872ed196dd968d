"""Phase stamps of one conv1x1_ring_kernel workgroup (diagnostic build: BD_LIB_NAME=libbasedet_r1x.so BD_EXTRA_FLAGS=-DBD_R1X_STAMP
python -m basedet_amd.build; run with BASEDET_HIP_LIB pointing at it).  Per wave and tile: cycles waiting for the ring (counted vmcnt), at
the step barrier, in DMA issue + fragment reads + MFMAs, waiting for the tile's epilogue operands, in the epilogue (request of the next
tile's operands, arithmetic, stores)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from micro_1x1_step import L, make, timeit

WANT = sys.argv[1:] or ["res3 conv3 fwd", "res4 conv1 fwd", "res4 conv3 fwd", "res5 conv1 fwd", "res4 conv1 dgrad"]
lib = ctypes.CDLL(_lib.LIB_PATH)
for tag, H, W, Cin, Cout, mode, add, mask, cnt in L:
    if not any(tag.startswith(w) for w in WANT):
        continue
    run, nbytes, _ = make(H, W, Cin, Cout, mode, add, mask, 1)
    us = timeit(run)
    out = (ctypes.c_ulonglong * 64)()
    assert lib.bd_debug_r1x_stamp(out) == 0
    print(f"== {tag}: {us:.1f} us per launch, {nbytes / us / 1e3:.0f} GB/s")
    for w in range(8):
        v = [out[w * 8 + k] for k in range(8)]
        rt, tiles, nsteps = v[7] >> 24, (v[7] >> 8) & 0xffff, v[7] & 0xff
        n = max(tiles, 1)
        print(f"  wave {w}: {tiles} tiles x {nsteps} steps; per tile: ring wait {v[1] / n:6.0f}  barrier {v[2] / n:6.0f}  reads+mfma {v[3] / n:6.0f}  operand wait {v[4] / n:6.0f}"
              f"  epilogue {v[5] / n:6.0f}  dma issue+loop {v[0] / n:5.0f} | total {v[6] / n:6.0f} cycles/tile, {rt * 10 / n:6.0f} ns/tile ({v[6] / max(rt, 1) / 10:.2f} GHz)")
    del run
    torch.cuda.empty_cache()
