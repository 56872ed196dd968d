"""Phase stamps of bd_bottleneck_fwd (diagnostic build: BD_LIB_NAME=libbd_stamp.so BD_EXTRA_FLAGS=-DBD_BN_STAMP python -m basedet_amd.build;
run with BASEDET_HIP_LIB=basedet_amd/lib/libbd_stamp.so).  s_memtime ticks at 100 MHz: 1 tick = 10 ns."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops, _lib
from tests.util import bf16_round, nchw_to_pm, pack_weights
for has_ds in (False, True):
    N, H, W = 16, 200, 336
    cin, ch, cout = (64 if has_ds else 256), 64, 256
    g = torch.Generator().manual_seed(1)
    xp = torch.randn(N * H * W, cin, generator=g).relu().to(torch.bfloat16).cuda()
    w1 = torch.randn(ch, cin, 1, 1, generator=g) * 0.1; w2 = torch.randn(ch, ch, 3, 3, generator=g) * 0.05
    w3 = torch.randn(cout, ch, 1, 1, generator=g) * 0.1; wd = torch.randn(cout, cin, 1, 1, generator=g) * 0.1
    (w1f, _), (w2f, _), (w3f, _), (wdf, _) = (pack_weights(ops, w) for w in (w1, w2, w3, wd))
    b = [torch.zeros(c, device="cuda") for c in (ch, ch, cout, cout)]
    y = torch.empty((N * H * W, cout), dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        ops.bottleneck_fwd(N, H, W, cin, ch, cout, xp, w1f, b[0], w2f, b[1], w3f, b[2], wdf if has_ds else None, b[3] if has_ds else None, y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.bottleneck_fwd(N, H, W, cin, ch, cout, xp, w1f, b[0], w2f, b[1], w3f, b[2], wdf if has_ds else None, b[3] if has_ds else None, y)
    e.record(); torch.cuda.synchronize()
    print(f"ds={has_ds}: {s.elapsed_time(e) / 10 * 1e3:.1f} us per launch")
    lib = _lib.load()
    if hasattr(lib, "bd_debug_bn_stamp"):
        out = (ctypes.c_ulonglong * 16)()
        lib.bd_debug_bn_stamp(out)
        t = [out[i] for i in range(6)]
        names = ["chunk loop (conv1)", "epilogue 1 + barrier", "conv2 MFMAs (+ res / warm-up issue)", "barrier + mid-2 write + barrier", "phase 3"]
        for i, n in enumerate(names):
            print(f"   {n:40s} {(t[i + 1] - t[i]) * 10} ns")
        print(f"   tile total {(t[5] - t[0]) * 10} ns")
