"""RoIAlign backward (fp32 atomic scatter) at C4's sizes: 16 images x 512 RoIs, 256 channels, P2..P5 of 800x1344 -- time with the RoIs of an
image in random slot order vs sorted by (pyramid level, y, x): does L2 / MALL locality of the read-modify-write traffic matter?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd import ops

N, C, rpi = 16, 256, 512
sizes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
STR = [4, 8, 16, 32, 64]
geom = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
rng = np.random.default_rng(3)

def boxes(n):
    # proposal-like: log-uniform sizes 24..500 px, aspect 0.5..2, centres uniform
    s = np.exp(rng.uniform(np.log(24), np.log(500), n)); a = np.exp(rng.uniform(np.log(0.5), np.log(2), n))
    w, h = s * np.sqrt(a), s / np.sqrt(a)
    cx, cy = rng.uniform(0, 1344, n), rng.uniform(0, 800, n)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clip(0, 1344); b[:, 1::2] = b[:, 1::2].clip(0, 800)
    return b.astype(np.float32)

def level(b):
    s = np.sqrt((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    return np.clip(np.floor(4 + np.log2(s / 224 + 1e-8)), 2, 5).astype(int)

rois = np.concatenate([boxes(rpi) for _ in range(N)], 0)
srt = rois.copy()
for n in range(N):
    b = rois[n * rpi:(n + 1) * rpi]
    key = np.lexsort((b[:, 0], b[:, 1] // 64, level(b)))
    srt[n * rpi:(n + 1) * rpi] = b[key]
labels = torch.ones(N * rpi, dtype=torch.int32, device="cuda")
gout = torch.randn(N * rpi, 49, C, device="cuda").to(torch.bfloat16)
gfeat = torch.zeros((N * geom.pix_per_img, C), dtype=torch.float32, device="cuda")
gpk = torch.zeros((N * geom.pix_per_img, C), dtype=torch.bfloat16, device="cuda")

def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3

for tag, r in (("random slot order", rois), ("sorted by level, y band, x", srt)):
    rd = torch.from_numpy(r).cuda()
    t32 = timeit(lambda: ops.roi_align_bwd(gout, geom, 4, STR, C, rd, labels, rpi, (7, 7), 2, gfeat))
    ws = torch.empty((ops.roi_align_bwd_bf16_workspace_bytes(geom, rpi),), dtype=torch.uint8, device="cuda")
    tg = timeit(lambda: ops.roi_align_bwd_bf16(gout, geom, 4, STR, C, rd, labels, rpi, (7, 7), 2, gpk, ws))
    tga = timeit(lambda: ops.roi_align_bwd_bf16(gout, geom, 4, STR, C, rd, labels, rpi, (7, 7), 2, gpk, ws, accumulate=True))
    tz = timeit(lambda: gfeat.zero_())
    tc = timeit(lambda: ops.f32_to_bf16(gfeat, gpk))
    print(f"{tag:30s}: fp32 scatter {t32:8.1f} us (+ zero fill {tz:6.1f} + convert {tc:6.1f})   deterministic tiles {tg:8.1f} us (accumulating: {tga:8.1f})", flush=True)
