"""Same-process A/B of the two nine-tap weight-gradient kernels (ring-staged conv_wgrad3x3_ring.hip vs register-staged conv_wgrad3x3.hip)
on the step's shapes: interleaved rounds, kernel + reduce per call (HIP events), and the rel-L2 between the two results.
usage: python scripts/micro_wgrad_ring.py [rounds=5]"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 16
PYR = ([100, 50, 25, 13, 7], [168, 84, 42, 21, 11])
SHAPES = [("head tower 256->256 P3..P7", PYR, 256, 256), ("cls_score 256->720 P3..P7", PYR, 256, 720),
          ("fpn out 256->256 100x168", ([100], [168]), 256, 256), ("fpn out 256->256 50x84", ([50], [84]), 256, 256),
          ("res4 conv2 256->256 50x84", ([50], [84]), 256, 256), ("res5 conv2 512->512 25x42", ([25], [42]), 512, 512),
          ("res3 conv2 128->128 100x168", ([100], [168]), 128, 128)]


def setup(hw, cin, cout):
    geo = ops.Geom(N, list(hw[0]), list(hw[1]))
    d = ops.conv_desc(geo, geo, cin, cout, 3, 3, 1, 1)
    x = torch.randn(geo.pixels, cin, device="cuda").to(torch.bfloat16)
    g = torch.randn(geo.pixels, cout, device="cuda").to(torch.bfloat16)
    ws = torch.empty(ops.conv2d_wgrad_bias_workspace_bytes(d) // 4 + 16, device="cuda")
    dw = torch.empty(cout, 3, 3, cin, device="cuda")
    db = torch.empty(cout, device="cuda")
    return geo, d, x, g, ws, dw, db


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


for name, hw, cin, cout in SHAPES:
    geo, d, x, g, ws, dw, db = setup(hw, cin, cout)
    fl = 2.0 * geo.pixels * cin * cout * 9
    res = {}
    t = {1: [], 5: []}
    for r in range(rounds):
        for knob in (1, 5):
            ops.set_route(wgrad=knob)
            t[knob].append(timed(lambda: ops.conv2d_wgrad_bias(d, x, g, dw, db, ws)))
            if r == 0:
                res[knob] = (dw.clone(), db.clone())
    ops.set_route(wgrad=1)
    rel = float((res[1][0] - res[5][0]).norm() / res[5][0].norm())
    relb = float((res[1][1] - res[5][1]).norm() / res[5][1].norm())
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print(f"{name:32s} ring {med[1]:7.1f} us {fl / med[1] / 1e6:7.1f} TF/s (min {min(t[1]):7.1f}) | staged {med[5]:7.1f} us {fl / med[5] / 1e6:7.1f} TF/s "
          f"(min {min(t[5]):7.1f}) | rel-L2 dw {rel:.2e} db {relb:.2e}", flush=True)
