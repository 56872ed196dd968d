"""Which tiles of bd_bottleneck_fwd differ from the separate launches (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd import ops
from tests.util import bf16_round, nchw_to_pm, pack_weights
N, H, W, has_ds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
cin, ch, cout = (64 if has_ds else 256), 64, 256
g = torch.Generator().manual_seed(1)
x = bf16_round(torch.randn(N, cin, H, W, generator=g).relu())
w1 = torch.randn(ch, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
w2 = torch.randn(ch, ch, 3, 3, generator=g) * (2.0 / (9 * ch)) ** 0.5
w3 = torch.randn(cout, ch, 1, 1, generator=g) * (1.0 / ch) ** 0.5
wd = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin) ** 0.5
b1, b2, b3, bd = (torch.randn(c, generator=g) * 0.2 for c in (ch, ch, cout, cout))
xp = nchw_to_pm(x)
(w1f, _), (w2f, _), (w3f, _), (wdf, _) = (pack_weights(ops, w) for w in (w1, w2, w3, wd))
b1d, b2d, b3d, bdd = (b.cuda() for b in (b1, b2, b3, bd))
geo = ops.single(N, H, W); M = N * H * W
m1 = torch.empty((M, ch), dtype=torch.bfloat16, device="cuda"); m2 = torch.empty_like(m1)
idt = torch.empty((M, cout), dtype=torch.bfloat16, device="cuda"); ref = torch.empty_like(idt)
ops.conv2d_fwd(ops.conv_desc(geo, geo, cin, ch, 1, 1, 1, 0), xp, w1f, b1d, m1, flags=ops.EPI_RELU)
ops.conv2d_fwd(ops.conv_desc(geo, geo, ch, ch, 3, 3, 1, 1), m1, w2f, b2d, m2, flags=ops.EPI_RELU)
if has_ds: ops.conv2d_fwd(ops.conv_desc(geo, geo, cin, cout, 1, 1, 1, 0), xp, wdf, bdd, idt)
ops.conv2d_fwd(ops.conv_desc(geo, geo, ch, cout, 1, 1, 1, 0), m2, w3f, b3d, ref, add=idt if has_ds else xp, flags=ops.EPI_RELU | ops.EPI_ADD_BEFORE)
y = torch.full((M, cout), -7.0, dtype=torch.bfloat16, device="cuda")
ops.bottleneck_fwd(N, H, W, cin, ch, cout, xp, w1f, b1d, w2f, b2d, w3f, b3d, wdf if has_ds else None, bdd if has_ds else None, y)
torch.cuda.synchronize()
err = ((y.float() - ref.float()).abs().amax(dim=1) > 0.05).cpu().numpy().reshape(N, H, W)
ty, tx = -(-H // 8), -(-W // 16)
total = N * ty * tx; per = -(-total // 8); grid = min(256, per * 8); stride = grid // 8
print("tiles", total, "per_xcd", per, "grid", grid)
bad = {}
for n in range(N):
    for a in range(ty):
        for b in range(tx):
            t = (n * ty + a) * tx + b
            e = err[n, a * 8:(a + 1) * 8, b * 16:(b + 1) * 16]
            xcd = t // per; k = (t - xcd * per) // stride
            bad.setdefault(k, []).append(float(e.mean()))
for k in sorted(bad):
    v = np.array(bad[k]); print("round", k, "tiles", len(v), "tiles with errors", int((v > 0).sum()), "mean bad-pixel fraction", float(v.mean()))
# row/col structure of errors inside bad tiles
rows = np.zeros(8); cols = np.zeros(16); cnt = 0
for n in range(N):
    for a in range(ty - 1):
        for b in range(tx - 1):
            e = err[n, a * 8:(a + 1) * 8, b * 16:(b + 1) * 16]
            if e.any(): rows += e.mean(1); cols += e.mean(0); cnt += 1
print("bad tiles", cnt, "row profile", np.round(rows / max(cnt, 1), 2), "col profile", np.round(cols / max(cnt, 1), 2))
