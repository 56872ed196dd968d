"""Same-process A/B of the two 1x1 weight-gradient kernels (ring-staged conv_wgrad1x1_ring.hip vs register-staged conv_wgrad1x1.hip) on
the step's shapes (R50 bottleneck conv1 / conv3 of layer2-4, FPN laterals): interleaved rounds, kernel + reduce per call (HIP events),
algorithmic GB/s (both operands once + the fp32 result), rel-L2 between the two results.   usage: python scripts/micro_wgrad1x1_ring.py [rounds=5]"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 16
SHAPES = [("layer2 conv1 512->128 100x168", 100, 168, 512, 128), ("layer2 conv3 128->512 100x168", 100, 168, 128, 512),
          ("layer3 conv1 1024->256 50x84", 50, 84, 1024, 256), ("layer3 conv3 256->1024 50x84", 50, 84, 256, 1024),
          ("layer4 conv1 2048->512 25x42", 25, 42, 2048, 512), ("layer4 conv3 512->2048 25x42", 25, 42, 512, 2048),
          ("lateral3 512->256 100x168", 100, 168, 512, 256), ("lateral4 1024->256 50x84", 50, 84, 1024, 256), ("lateral5 2048->256 25x42", 25, 42, 2048, 256)]


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, h, w, cin, cout in SHAPES:
    geo = ops.single(N, h, w)
    d = ops.conv_desc(geo, geo, cin, cout, 1, 1, 1, 0)
    x = torch.randn(geo.pixels, cin, device="cuda").to(torch.bfloat16)
    g = torch.randn(geo.pixels, cout, device="cuda").to(torch.bfloat16)
    ws = torch.empty(ops.conv2d_wgrad_workspace_bytes(d) // 4 + 16, device="cuda")
    dw = torch.empty(cout, 1, 1, cin, device="cuda")
    nbytes = 2.0 * geo.pixels * (cin + cout) + 4.0 * cin * cout
    res, t = {}, {1: [], 5: []}
    for r in range(rounds):
        for knob in (1, 5):
            ops.set_route(wgrad=knob)
            t[knob].append(timed(lambda: ops.conv2d_wgrad(d, x, g, dw, ws)))
            if r == 0:
                res[knob] = dw.clone()
    ops.set_route(wgrad=1)
    rel = float((res[1] - res[5]).norm() / res[5].norm())
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print(f"{name:32s} ring {med[1]:7.1f} us {nbytes / med[1] / 1e3:7.0f} GB/s (min {min(t[1]):7.1f}) | staged {med[5]:7.1f} us {nbytes / med[5] / 1e3:7.0f} GB/s "
          f"(min {min(t[5]):7.1f}) | rel-L2 {rel:.2e}", flush=True)
