"""1x1 data gradients with their epilogue operands (ReLU mask + accumulated gradient), as the bottleneck blocks issue them:
time per launch and algorithmic HBM rate.  BD_KNOB = bd_conv_desc.route[1] mask, BD_PRE_KSTEPS = prefetch threshold."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops
if os.environ.get("BD_KNOB"):
    ops.set_route(patch3x3=int(os.environ["BD_KNOB"]))


def bench(N, H, W, Cin, Cout, epi, iters=20):
    gin = ops.single(N, H, W); gout = gin.conv_out(1, 1, 0)
    d = ops.conv_desc(gin, gout, Cin, Cout, 1, 1, 1, 0)
    g = torch.randn(gout.pixels, Cout, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(Cin, 1, Cout, device="cuda") * 0.02).to(torch.bfloat16)
    dx = torch.empty(gin.pixels, Cin, device="cuda", dtype=torch.bfloat16)
    add = torch.randn_like(dx) if "a" in epi else None
    mask = torch.randn_like(dx) if "m" in epi else None
    flags = (ops.EPI_ADD_AFTER if add is not None else 0) | (ops.EPI_MASK if mask is not None else 0)
    def run():
        ops.conv2d_dgrad(d, g, wt, dx, add=add, mask=mask, flags=flags)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    nb = 2.0 * (gout.pixels * Cout + gin.pixels * Cin * (1 + (add is not None) + (mask is not None))) + 2.0 * Cin * Cout
    print(f"dgrad {H}x{W} Cin={Cin:5d} Cout={Cout:5d} epi={epi or '-':3s}: {ms*1e3:7.1f} us {2.0*gout.pixels*Cin*Cout/ms/1e9:6.0f} TF/s {nb/ms/1e6:6.0f} GB/s", flush=True)


for (h, w, cin, cout) in ((50, 84, 1024, 256), (100, 168, 512, 128), (100, 168, 512, 256), (25, 42, 2048, 512), (50, 84, 256, 1024)):
    for epi in ("", "a", "m", "am"):
        bench(16, h, w, cin, cout, epi)
