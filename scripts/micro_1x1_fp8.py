"""1x1 forward launches of one step: the bf16 dense kernel vs the fp8 kernels on the same shapes (e4m3 input twin resident, as in the model).
python scripts/micro_1x1_fp8.py"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops

N = 16
L = [
    ("res3 conv1 fwd 512->128", 100, 168, 512, 128, 0),
    ("res3 conv3 fwd 128->512 +add", 100, 168, 128, 512, 1),
    ("res4 conv1 fwd 1024->256", 50, 84, 1024, 256, 0),
    ("res4 conv3 fwd 256->1024 +add", 50, 84, 256, 1024, 1),
    ("res5 conv1 fwd 2048->512", 25, 42, 2048, 512, 0),
    ("res5 conv3 fwd 512->2048 +add", 25, 42, 512, 2048, 1),
]


def timeit(run, iters=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for tag, H, W, Cin, Cout, add in L:
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
    M = geo.pixels
    x = torch.randn(M, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, 1, Cin, device="cuda") * 0.03)
    wb = w.to(torch.bfloat16)
    y = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    addt = torch.randn(M, Cout, device="cuda").to(torch.bfloat16) if add else None
    bias = torch.zeros(Cout, device="cuda")
    fl = ops.EPI_RELU | (ops.EPI_ADD_BEFORE if add else 0)
    x8 = torch.empty(M * Cin, dtype=torch.uint8, device="cuda")
    ops.quantize_fp8(x, 1.0, x8)
    wq = torch.empty((Cout, 1, Cin), dtype=torch.uint8, device="cuda")
    ws = torch.empty((Cout,), dtype=torch.float32, device="cuda")
    ops.weight_pack_fp8(w.contiguous(), None, Cout, 1, Cin, 1.0, wq, ws)
    y8 = torch.empty(M * Cout, dtype=torch.uint8, device="cuda")
    t16 = timeit(lambda: ops.conv2d_fwd(d, x, wb, bias, y, add=addt, flags=fl))
    t8 = timeit(lambda: ops.conv2d_fwd_fp8(d, x8, wq, ws, bias, y, add=addt, flags=fl))
    t8t = timeit(lambda: ops.conv2d_fwd_fp8(d, x8, wq, ws, bias, y, add=addt, flags=fl, y8=y8, q_scale=1.0))
    yb = torch.empty((Cout // 32, M), dtype=torch.int32, device="cuda")
    tn = timeit(lambda: ops.conv1x1_fp8(d, 0, x8, wq, ws, bias, y, add=addt, flags=fl))
    tnt = timeit(lambda: ops.conv1x1_fp8(d, 0, x8, wq, ws, bias, y, add=addt, flags=fl, bits=yb, y8=y8))
    print(f"{tag:34s} bf16 {t16:7.1f} us   generic fp8 {t8:7.1f} / + twin {t8t:7.1f} us   dense fp8 {tn:7.1f} / + bits + twin {tnt:7.1f} us", flush=True)
