"""Phase stamps of one conv_wgrad3x3_ring_kernel workgroup (diagnostic build: BD_LIB_NAME=libbasedet_rk.so BD_EXTRA_FLAGS=-DBD_RK_STAMP
python -m basedet_amd.build; run with BASEDET_HIP_LIB pointing at it).  Per wave and step (both phases summed): cycles in the load
segments (fragment reads, DMA requests, counted vmcnt), at the barrier behind them, in the MFMA segments, at the barrier behind those."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops, _lib

N = 16
for tag, Cin, Cout, sizes in (("head tower 256->256, 5 levels", 256, 256, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]),
                              ("res4 conv2 256->256 @50x84", 256, 256, [(50, 84)])):
    geo = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
    d = ops.conv_desc(geo, geo, Cin, Cout, 3, 3, 1, 1)
    x = torch.randn(geo.pixels, Cin, device="cuda").to(torch.bfloat16)
    g = (torch.randn(geo.pixels, Cout, device="cuda") * 1e-3).to(torch.bfloat16)
    dw = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device="cuda")
    ws = torch.empty((ops.conv2d_wgrad_workspace_bytes(d) // 4 + 64,), dtype=torch.float32, device="cuda")
    for _ in range(5):
        ops.conv2d_wgrad(d, x, g, dw, ws)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv2d_wgrad(d, x, g, dw, ws)
    e.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    L = ctypes.CDLL(_lib.LIB_PATH)
    assert L.bd_debug_rk_stamp(out) == 0
    print(f"{tag}: {s.elapsed_time(e) * 100:.1f} us per launch (kernel + reduce)")
    for w in range(8):
        v = [out[w * 8 + k] for k in range(7)]
        n = max(v[4], 1)
        print(f"  wave {w}: steps {v[4]}  per step (two phases): load segment {v[0] / n:6.0f}  barrier {v[1] / n:6.0f}  mfma segment {v[2] / n:6.0f}  barrier {v[3] / n:6.0f}"
              f"  sum {sum(v[:4]) / n:6.0f}   (loop total {v[5]} cycles in {v[6] * 10} ns: {v[5] / max(v[6], 1) / 10:.2f} GHz)")
