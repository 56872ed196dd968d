"""Fixed per-tile cost of conv1x1_dense_kernel: M = 268 800 pixels, Cout = 512, Cin = 32 .. 512 (1 .. 16 K steps), forward with residual +
ReLU + gate bits; time ~ tiles x (F + steps x S).   python scripts/exp/dense_overhead.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops

N, H, W, Co = 16, 100, 168, 512
geo = ops.single(N, H, W)
M = geo.pixels
res = []
for C in (32, 64, 128, 256, 512):
    d = ops.conv_desc(geo, geo, C, Co, 1, 1, 1, 0)
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 1, C, device="cuda") * 0.03).to(torch.bfloat16)
    y = torch.empty(M, Co, device="cuda", dtype=torch.bfloat16)
    add = torch.randn(M, Co, device="cuda").to(torch.bfloat16)
    bits = torch.empty((Co // 32, M), device="cuda", dtype=torch.int32)
    bias = torch.zeros(Co, device="cuda")
    out = []
    for variant in ("plain", "add+relu+bits"):
        def run():
            if variant == "plain":
                ops.conv2d_fwd(d, x, w, bias, y, flags=0)
            else:
                ops.conv2d_fwd(d, x, w, bias, y, add=add, flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE, bits=bits)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / 20 * 1e3)
    res.append((C // 32, out))
    print(f"Cin {C:4d} ({C // 32:2d} K steps): plain {out[0]:7.1f} us   add+relu+bits {out[1]:7.1f} us", flush=True)
for v in (0, 1):
    (k0, t0), (k1, t1) = (res[0][0], res[0][1][v]), (res[-1][0], res[-1][1][v])
    S = (t1 - t0) / (k1 - k0)
    print(f"variant {v}: {S:.2f} us per K step over the launch; fixed part {t0 - k0 * S:.1f} us of the launch (HBM floor of the epilogue bytes: "
          f"{(2.0 * M * Co * (1 + v) + (M * Co / 8 if v else 0)) / 5.5e6:.1f} us at 5.5 TB/s)")
