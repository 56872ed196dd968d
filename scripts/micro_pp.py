"""Staggered 256-channel patch kernel (default; bd_conv_set_patch3x3 bit 6 disables it) against the 128-channel patch kernel: max difference + timing."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
import torch
from basedet_amd import ops
from micro_conv import bench


def check(N, H, W, Cin, Cout, mode, flags=0):
    gin = ops.single(N, H, W); gout = gin.conv_out(3, 1, 1)
    d = ops.conv_desc(gin, gout, Cin, Cout, 3, 3, 1, 1)
    torch.manual_seed(0)
    ck, co = (Cin, Cout) if mode == "fwd" else (Cout, Cin)
    x = torch.randn(gin.pixels, ck, device="cuda").to(torch.bfloat16)
    w = (torch.randn(co, 9, ck, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(co, device="cuda") if mode == "fwd" else None
    add = torch.randn(gin.pixels, co, device="cuda").to(torch.bfloat16)
    outs = []
    for knob in (7 | 64 | 512, 7 | 64 | 256):
        ops.L().bd_conv_set_patch3x3(knob)
        y = torch.full((gin.pixels, co), 7.0, device="cuda", dtype=torch.bfloat16)
        if mode == "fwd":
            ops.conv2d_fwd(d, x, w, b, y, add=add if flags & ops.EPI_ADD_BEFORE else None, flags=flags)
        else:
            ops.conv2d_dgrad(d, x, w, y, add=add if flags else None, mask=add if flags else None, flags=flags)
        torch.cuda.synchronize()
        outs.append(y.float())
    diff = (outs[0] - outs[1]).abs().max().item()
    print(f"check {mode} N={N} {H}x{W} {Cin}->{Cout} flags={flags}: max|diff|={diff:.4g} ref max={outs[0].abs().max().item():.3g}", flush=True)
    return diff


bad = 0
for (N, H, W, Cin, Cout) in ((2, 13, 21, 256, 256), (1, 7, 11, 64, 256), (2, 25, 42, 512, 512), (1, 50, 84, 256, 720), (3, 17, 33, 128, 264)):
    bad += check(N, H, W, Cin, Cout, "fwd", ops.EPI_RELU) > 0.05
    bad += check(N, H, W, Cin, Cout, "fwd", ops.EPI_ADD_BEFORE | ops.EPI_RELU) > 0.05
for (N, H, W, Cin, Cout) in ((2, 13, 21, 256, 256), (2, 25, 42, 512, 512), (1, 20, 30, 264, 128)):
    bad += check(N, H, W, Cin, Cout, "dgrad") > 0.05
    bad += check(N, H, W, Cin, Cout, "dgrad", ops.EPI_ADD_BEFORE | ops.EPI_MASK) > 0.05
print("MISMATCHES", bad, flush=True)
for rep in range(2):
    for knob in (3 | 64 | 512, 3 | 64 | 256, 3):
        ops.L().bd_conv_set_patch3x3(knob)
        print("knob", knob, flush=True)
        for mode in ("fwd", "dgrad"):
            for (h, w, cin, cout) in ((100, 168, 256, 256), (50, 84, 256, 256), (25, 42, 512, 512), (100, 168, 128, 128), (200, 336, 64, 64)):
                bench(16, h, w, cin, cout, mode=mode)
