"""3x3 weight-gradient launches of the RetinaNet step: bf16 nine-tap kernel vs the one-byte kernel on the same shapes (twins resident).
python scripts/micro_wgrad_fp8.py"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops

N = 16
SH = [("head tower 256->256, 5 levels", 256, 256, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]),
      ("cls_score 256->720, 5 levels", 256, 720, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]),
      ("fpn output 256->256 @100x168", 256, 256, [(100, 168)]),
      ("res4 conv2 256->256 @50x84", 256, 256, [(50, 84)]),
      ("res5 conv2 512->512 @25x42", 512, 512, [(25, 42)])]


def timeit(run, iters=10):
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


for tag, Cin, Cout, sizes in SH:
    geo = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
    d = ops.conv_desc(geo, geo, Cin, Cout, 3, 3, 1, 1)
    x = torch.randn(geo.pixels, Cin, device="cuda").to(torch.bfloat16)
    g = (torch.randn(geo.pixels, Cout, device="cuda") * 1e-3).to(torch.bfloat16)
    x8 = torch.empty(x.numel(), dtype=torch.uint8, device="cuda"); ops.quantize_fp8(x, 1.0, x8)
    g8 = torch.empty(g.numel(), dtype=torch.uint8, device="cuda"); ops.quantize_bf8(g, 4096.0, g8)
    dw = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device="cuda")
    ws = torch.empty((max(ops.conv2d_wgrad_workspace_bytes(d), ops.conv2d_wgrad_fp8_workspace_bytes(d)) // 4 + 64,), dtype=torch.float32, device="cuda")
    t16 = timeit(lambda: ops.conv2d_wgrad(d, x, g, dw, ws))
    t8 = timeit(lambda: ops.conv2d_wgrad_fp8(d, x8, g8, 1.0 / 4096.0, dw, ws))
    fl = 2.0 * geo.pixels * Cin * Cout * 9
    print(f"{tag:34s} bf16 {t16:7.1f} us ({fl / t16 / 1e6:6.0f} TF/s)   fp8 {t8:7.1f} us ({fl / t8 / 1e6:6.0f} TF/s)", flush=True)
