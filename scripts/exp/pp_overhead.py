"""Fixed per-workgroup cost (prologue + epilogue) of conv3x3_pp_kernel: the head geometry with Cin = 64 .. 512 (1 .. 8 K blocks), Cout = 256;
time = rounds x (P + E + kblocks x B).   python scripts/exp/pp_overhead.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops

N, Co = 16, 256
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
geo = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
res = []
for C in (64, 128, 256, 512):
    d = ops.conv_desc(geo, geo, C, Co, 3, 3, 1, 1)
    x = torch.randn(geo.pixels, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9, C, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(geo.pixels, Co, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(Co, device="cuda")
    for _ in range(5):
        ops.conv2d_fwd(d, x, w, bias, y, flags=ops.EPI_RELU)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        ops.conv2d_fwd(d, x, w, bias, y, flags=ops.EPI_RELU)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    res.append((C // 64, us))
    print(f"Cin {C:4d} ({C // 64} K blocks): {us:7.1f} us per launch = {us / 6:6.1f} us per round of 256 workgroups")
(k0, t0), (k1, t1) = res[0], res[-1]
B = (t1 - t0) / (k1 - k0) / 6
print(f"per K block {B:.2f} us per workgroup; fixed part (prologue + epilogue) {t0 / 6 - k0 * B:.2f} us per workgroup")
