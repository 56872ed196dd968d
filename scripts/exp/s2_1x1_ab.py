"""1x1 / stride-2 shortcut launches: the dense kernel with strided row mapping vs the generic kernel (bd_conv_desc.route[1] bit 12), forward and
the in-place sparse data gradient, same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops

def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

N = 16
for (h, w, cin, cout) in ((200, 336, 256, 512), (100, 168, 512, 1024), (50, 84, 1024, 2048)):
    gin = ops.single(N, h, w); gout = gin.conv_out(1, 2, 0)
    d = ops.conv_desc(gin, gout, cin, cout, 1, 1, 2, 0)
    x = torch.randn(gin.pixels, cin, device="cuda").to(torch.bfloat16)
    wf = (torch.randn(cout, 1, cin, device="cuda") * 0.02).to(torch.bfloat16)
    wt = (torch.randn(cin, 1, cout, device="cuda") * 0.02).to(torch.bfloat16)
    y = torch.empty(gout.pixels, cout, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(gout.pixels, cout, device="cuda").to(torch.bfloat16)
    dx = torch.randn(gin.pixels, cin, device="cuda").to(torch.bfloat16)
    act = torch.relu(torch.randn(gin.pixels, cin, device="cuda")).to(torch.bfloat16)
    res = []
    for knob in (1 | 4096, 1):
        ops.set_route(patch3x3=knob)
        tf = timeit(lambda: ops.conv2d_fwd(d, x, wf, None, y))
        td = timeit(lambda: ops.conv2d_dgrad(d, g, wt, dx, add=dx, mask=act, flags=ops.EPI_ADD_BEFORE | ops.EPI_MASK | ops.EPI_SPARSE))
        res.append((tf, td))
    ops.set_route(patch3x3=1)
    print(f"{h}x{w} {cin}->{cout} s2: fwd generic {res[0][0]:7.1f} us -> dense {res[1][0]:7.1f} us   sparse dgrad generic {res[0][1]:7.1f} us -> dense {res[1][1]:7.1f} us", flush=True)
