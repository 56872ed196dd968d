"""Per-level relative L2 error of bd_roi_align_bwd_pk (packed-bf16 atomics) against the float64 adjoint at C4's sizes; BD_ROI_BWD_SEP=0/1 selects the
per-bin / the separable per-RoI scatter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_rcnn_ops_gpu import _ops, _geom, _rand_boxes, _bf16, _dev, STRIDES, orc
ops = _ops()
for seed in (11, 12):
    rng = np.random.default_rng(seed)
    N, C, rpi = 2, 256, 512
    sizes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    nlev = 4
    geom = _geom(N, sizes); ppi = geom.pix_per_img
    rois = np.concatenate([_rand_boxes(rng, rpi, 1344, 800, 8, 700) for _ in range(N)], 0)
    labels = np.ones(N * rpi, np.int32)
    bidx = np.repeat(np.arange(N), rpi)
    gout = _bf16(rng.normal(0, 1, (N * rpi, 49, C)).astype(np.float32))
    refg = orc.roi_align_backward(gout.float().numpy(), [(N, h, w, C) for h, w in sizes[:nlev]], rois, bidx, STRIDES[:nlev], 7, 7, 2)
    gpk = torch.zeros((N * ppi, C), dtype=torch.bfloat16, device="cuda")
    ops.roi_align_bwd_pk(gout.cuda(), geom, nlev, STRIDES, C, _dev(rois), _dev(labels), rpi, (7, 7), 2, gpk)
    gk = gpk.float().cpu().numpy().reshape(N, ppi, C)
    o = 0; out = []
    for l, (h, w) in enumerate(sizes[:nlev]):
        ref_l = refg[l].reshape(N, h * w, C)
        out.append(np.linalg.norm(gk[:, o:o + h * w] - ref_l) / np.linalg.norm(ref_l)); o += h * w
    print("SEP", os.environ.get("BD_ROI_BWD_SEP", "1"), "seed", seed, ["%.4f" % e for e in out])
