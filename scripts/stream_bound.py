"""How close are the HBM-bound 1x1 layers to a plain streaming kernel moving the same bytes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basedet_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

import os as _os
ops.set_route(patch3x3=int(_os.environ.get("BD_PATCH3X3", "3")))
for (H, W, Cin, Cout, with_add) in ((200, 336, 64, 256, True), (100, 168, 128, 512, True), (50, 84, 256, 1024, True), (50, 84, 1024, 256, False),
                                    (100, 168, 512, 128, False), (200, 336, 256, 64, False)):
    N = 16
    g = ops.single(N, H, W); d = ops.conv_desc(g, g, Cin, Cout, 1, 1, 1, 0)
    x = torch.randn(g.pixels, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Cout, 1, Cin, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(g.pixels, Cout, device="cuda", dtype=torch.bfloat16)
    add = torch.randn(g.pixels, Cout, device="cuda").to(torch.bfloat16) if with_add else None
    bias = torch.zeros(Cout, device="cuda")
    t_conv = timeit(lambda: ops.conv2d_fwd(d, x, w, bias, y, add=add, flags=ops.EPI_RELU | (ops.EPI_ADD_BEFORE if with_add else 0)))
    nbytes = 2 * g.pixels * (Cin + Cout * (2 if with_add else 1))
    # streaming proxies moving the same bytes: copy of the input + (add or copy) of the output-sized tensors
    xs = torch.empty_like(x)
    if with_add:
        t_stream = timeit(lambda: (xs.copy_(x), torch.add(add, add, out=y)))      # reads x, writes xs (extra), reads add twice (1 from L2), writes y
        t_stream2 = timeit(lambda: torch.add(add, add, out=y))
    else:
        t_stream = timeit(lambda: (xs.copy_(x), y.copy_(y)))
        t_stream2 = timeit(lambda: y.copy_(y))
    print(f"{H}x{W} {Cin}->{Cout} add={with_add}: conv {t_conv*1e3:7.1f} us = {nbytes/t_conv/1e6:6.0f} GB/s | torch stream proxies {t_stream*1e3:7.1f} / {t_stream2*1e3:7.1f} us", flush=True)
