"""Race screen of bd_bottleneck_fwd: the same launch repeated under a competing memory stream; every output must equal the first bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops
from tests.util import pack_weights
for has_ds in (False, True):
    N, H, W = 16, 200, 336
    cin, ch, cout = (64 if has_ds else 256), 64, 256
    g = torch.Generator().manual_seed(1)
    xp = torch.randn(N * H * W, cin, generator=g).relu().to(torch.bfloat16).cuda()
    w1 = torch.randn(ch, cin, 1, 1, generator=g) * 0.1; w2 = torch.randn(ch, ch, 3, 3, generator=g) * 0.05
    w3 = torch.randn(cout, ch, 1, 1, generator=g) * 0.1; wd = torch.randn(cout, cin, 1, 1, generator=g) * 0.1
    (w1f, _), (w2f, _), (w3f, _), (wdf, _) = (pack_weights(ops, w) for w in (w1, w2, w3, wd))
    b = [torch.randn(c, generator=g).cuda() * 0.1 for c in (ch, ch, cout, cout)]
    def run(y):
        ops.bottleneck_fwd(N, H, W, cin, ch, cout, xp, w1f, b[0], w2f, b[1], w3f, b[2], wdf if has_ds else None, b[3] if has_ds else None, y)
    ref = torch.empty((N * H * W, cout), dtype=torch.bfloat16, device="cuda"); run(ref); torch.cuda.synchronize()
    side = torch.cuda.Stream(); junk = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    bad = 0
    y = torch.empty_like(ref)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
        with torch.cuda.stream(side):
            junk.add_(1)                                   # a competing memory stream, as the weight-gradient stream is in the step
        y.fill_(-1.0)
        run(y)
        if not torch.equal(y, ref):
            bad += 1
            d = (y.float() - ref.float()).abs()
            print("ds", has_ds, "iteration", it, "differs:", int((d > 0).sum()), "elements, max", float(d.max()), "nan", int(torch.isnan(y.float()).sum()), flush=True)
            if bad > 5: break
    torch.cuda.synchronize()
    print("ds", has_ds, "mismatching iterations:", bad)
