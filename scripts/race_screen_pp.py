"""Race screen for the staggered patch kernel (counted vmcnt + raw barriers): many launches per shape, each compared bit for bit with
the 128-channel instance (same accumulation order).  Any difference = an ordering bug in the DMA ring / stagger."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops

shapes = [(16, 200, 336, 64, 64), (16, 100, 168, 256, 40), (16, 100, 168, 256, 256), (4, 100, 168, 64, 720), (16, 50, 84, 256, 256), (16, 25, 42, 512, 512), (3, 37, 53, 200, 264), (16, 100, 168, 720, 256),
          (2, 13, 21, 256, 256), (16, 100, 168, 40, 256)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for (N, H, W, Cin, Cout) in shapes:
    gin = ops.single(N, H, W)
    d = ops.conv_desc(gin, gin, Cin, Cout, 3, 3, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(N + H + Cin)
    x = torch.randn(gin.pixels, Cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(Cout, 9, Cin, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda", generator=g)
    ops.set_route(patch3x3=3 | 64 | 512)
    ref = torch.empty((gin.pixels, Cout), device="cuda", dtype=torch.bfloat16)
    ops.conv2d_fwd(d, x, w, b, ref, flags=ops.EPI_RELU)
    ops.set_route(patch3x3=3)
    # a competing stream keeps the memory system busy (DMA latencies vary)
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    nbad = 0
    for it in range(iters):
        with torch.cuda.stream(side):
            junk.add_(1)
        y = torch.full((gin.pixels, Cout), -1.0, device="cuda", dtype=torch.bfloat16)
        ops.conv2d_fwd(d, x, w, b, y, flags=ops.EPI_RELU)
        if not torch.equal(y, ref):
            nbad += 1
    torch.cuda.synchronize()
    print(f"N={N} {H}x{W} {Cin}->{Cout}: {nbad} / {iters} launches differ", flush=True)
    bad += nbad
ops.set_route(patch3x3=3)
print("RACE SCREEN", "FAILED" if bad else "clean", flush=True)
sys.exit(1 if bad else 0)
