"""conv3x3_pp8_kernel per shape: the 3x3 / stride-1 launches of RetinaNet-R101 at batch 32 (res3 / res4 / res5 bodies, the five-level head), forward and
data gradient, one-byte operands, with the bf16 conv3x3_pp_kernel launch of the same shape beside it.
   [BASEDET_HIP_LIB=...] python scripts/exp/pp8_shapes.py [iters=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = int(os.environ.get("PP8_N", 32))          # images (PP8_N=24: res5 is one grid round of 252 tiles)
ONLY = os.environ.get("PP8_ONLY", "")     # substring of the shape names to run
HEAD = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
SHAPES = [("res3 128", [(100, 168)], 128, 128), ("res4 256", [(50, 84)], 256, 256), ("res5 512", [(25, 42)], 512, 512),
          ("head 256", HEAD, 256, 256), ("head cls 720", HEAD, 256, 720)]


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS * 1e3


for name, sizes, C, CO in SHAPES:
    if ONLY and ONLY not in name: continue
    geo = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
    fl = 2.0 * geo.pixels * C * CO * 9
    x = torch.randn(geo.pixels, C, device="cuda").to(torch.bfloat16)
    gy = (torch.randn(geo.pixels, CO, device="cuda") * 2.0 ** -10).to(torch.bfloat16)
    w = (torch.randn(CO, 9, C, device="cuda") * 0.02).contiguous()
    wb, wtb = w.to(torch.bfloat16), w.permute(2, 1, 0).contiguous().to(torch.bfloat16)      # [CO][9][C], [C][9][CO]
    bias = torch.randn(CO, device="cuda")
    d = ops.conv_desc(geo, geo, C, CO, 3, 3, 1, 1)
    y = torch.empty(geo.pixels, CO, device="cuda", dtype=torch.bfloat16)
    y8 = torch.empty(geo.pixels, CO, device="cuda", dtype=torch.uint8)
    dx = torch.empty(geo.pixels, C, device="cuda", dtype=torch.bfloat16)
    dx8 = torch.empty(geo.pixels, C, device="cuda", dtype=torch.uint8)
    xq = torch.empty(x.numel(), dtype=torch.uint8, device="cuda"); ops.quantize_fp8(x, 1.0, xq)
    g8 = torch.empty(gy.numel(), dtype=torch.uint8, device="cuda"); ops.quantize_bf8(gy, 1024.0, g8)
    wq = torch.empty((CO, 9, C), dtype=torch.uint8, device="cuda"); ws = torch.empty(CO, device="cuda")
    ops.weight_pack_fp8(w, None, CO, 9, C, 1.0, wq, ws)
    wqt = torch.empty((C, 9, CO), dtype=torch.uint8, device="cuda"); wst = torch.empty(C, device="cuda")
    ops.weight_pack_fp8_t(w, None, CO, 9, C, 1024.0, wqt, wst)
    rows = []
    for label, fn in (("fwd  bf16", lambda: ops.conv2d_fwd(d, x, wb, bias, y, flags=ops.EPI_RELU)),
                      ("fwd  fp8 ", lambda: ops.conv2d_fwd_fp8(d, xq, wq, ws, bias, y, flags=ops.EPI_RELU)),
                      ("fwd  fp8+twin", lambda: ops.conv2d_fwd_fp8(d, xq, wq, ws, bias, y, flags=ops.EPI_RELU, y8=y8, q_scale=1.0)),
                      ("dgrad bf16", lambda: ops.conv2d_dgrad(d, gy, wtb, dx)),
                      ("dgrad fp8 ", lambda: ops.conv2d_dgrad_fp8(d, g8, wqt, wst, dx, q_scale=1024.0)),
                      ("dgrad fp8+twin", lambda: ops.conv2d_dgrad_fp8(d, g8, wqt, wst, dx, dx8=dx8, q_scale=1024.0))):
        try:
            us = timeit(fn)
        except Exception as ex:          # a shape one of the libraries does not take
            print(f"{name:13s} {label:15s} -- {type(ex).__name__}: {ex}"); continue
        print(f"{name:13s} {label:15s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s   {ops.L().bd_conv_last_kernel().decode()}", flush=True)
