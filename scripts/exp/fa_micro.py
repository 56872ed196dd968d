"""FreeAnchor loss kernels alone at C2-like sizes (16 images, 201 600 anchors, 80 classes): for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from basedet_amd import ops
from oracle import box_ops
N, K, apix, ld = 16, 80, 9, 40
G = int(os.environ.get("FA_G", "15"))
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
strides = [8, 16, 32, 64, 128]
scales = [[s * 4, s * 4 * 2 ** (1 / 3), s * 4 * 2 ** (2 / 3)] for s in strides]
anchors = np.concatenate(box_ops.default_anchors(sizes, strides, scales, [[0.5, 1, 2]], 0.5), 0).astype(np.float32)
A = anchors.shape[0]
rng = np.random.default_rng(0)
gt = np.zeros((N, G, 5), np.float32)
cx, cy = rng.uniform(100, 1244, (N, G)), rng.uniform(100, 700, (N, G))
w, h = rng.uniform(30, 400, (N, G)), rng.uniform(30, 400, (N, G))
gt[..., 0] = np.clip(cx - w / 2, 0, 1344); gt[..., 1] = np.clip(cy - h / 2, 0, 800); gt[..., 2] = np.clip(cx + w / 2, 0, 1344); gt[..., 3] = np.clip(cy + h / 2, 0, 800)
gt[..., 4] = rng.integers(1, K + 1, (N, G))
num = np.full((N,), G, np.int32)
dev = "cuda"
lg = (torch.randn(N * A, K, device=dev) * 1.5 - 2).to(torch.bfloat16)
off = (torch.randn(N * (A // apix), ld, device=dev) * 0.05).to(torch.bfloat16)
d_lg, d_of = torch.empty_like(lg), torch.empty_like(off)
loss = torch.zeros(2, device=dev)
ws = torch.empty(ops.freeanchor_workspace_bytes(N, G, 50, A), dtype=torch.uint8, device=dev)
an, gtd, nd = torch.from_numpy(anchors).to(dev), torch.from_numpy(gt).to(dev), torch.from_numpy(num).to(dev)
for _ in range(5):
    ops.freeanchor_loss_fwd_bwd(lg, off, ld, apix, an, K, gtd, nd, (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), 0.6, 50, 0.11, 0.75, 0.5, 2.0, loss, d_lg, d_of, ws)
torch.cuda.synchronize()
print("A", A, "loss", loss.cpu().numpy())
