"""roi_pool / assign_rois (basedet/layers/common/roi_pool.py:12-78) and sample_labels (layers/common/sampling.py:7-30) on the HIP
kernels, with the reference's signatures: NCHW fp32 feature list in, (R, C, PH, PW) out."""
import math
from typing import List

import torch

from .. import ops
from .._lib import check, i32arr, ptr, stream_ptr


def _pair(v):
    return (int(v), int(v)) if isinstance(v, (int, float)) else (int(v[0]), int(v[1]))


def assign_rois(rois, strides):
    """roi_pool.py:12-32: (rois + one dummy row per level, assigned level index + arange(levels))."""
    rois = rois.detach().float().contiguous()
    min_level, max_level = int(math.log2(strides[0])), int(math.log2(strides[-1]))
    R = rois.shape[0]
    lv = torch.empty((R,), dtype=torch.int32, device=rois.device)
    check(ops.L().bd_assign_roi_levels(ptr(rois), rois.shape[1], R, min_level, max_level, ptr(lv), stream_ptr()), "bd_assign_roi_levels")
    n = len(strides)
    lv = torch.cat([lv, torch.arange(n, dtype=torch.int32, device=rois.device)])
    rois = torch.cat([rois, torch.zeros((n, rois.shape[-1]), dtype=rois.dtype, device=rois.device)])
    return rois, lv


def roi_pool(features: List[torch.Tensor], rois: torch.Tensor, strides: List[int], pool_shape, pooler_type: str = "roi_align"):
    """roi_pool.py:35-78.  features: NCHW tensors, one per stride; rois (R, 5) = (batch index, x1, y1, x2, y2).
    "roi_align" = average, 2x2 samples, aligned (bd_roi_align_fwd: bf16 channel-last pyramid, level chosen in the kernel);
    "roi_pool" = max pooling (bd_roi_pool_max_fwd on fp32, per level)."""
    assert pooler_type in ("roi_align", "roi_pool")
    assert len(strides) == len(features)
    PH, PW = _pair(pool_shape)
    dev = features[0].device
    N, Cn = features[0].shape[0], features[0].shape[1]
    R = rois.shape[0]
    rois = rois.detach().float().contiguous()
    if pooler_type == "roi_pool":
        _, lv = assign_rois(rois, strides)
        lv = lv[:R]
        out = torch.zeros((R, Cn, PH, PW), dtype=torch.float32, device=dev)
        for i, (f, s) in enumerate(zip(features, strides)):
            sel = torch.nonzero(lv == i).flatten()
            if sel.numel() == 0:
                continue
            f = f.float().contiguous()
            o = torch.empty((sel.numel(), Cn, PH, PW), dtype=torch.float32, device=dev)
            check(ops.L().bd_roi_pool_max_fwd(ptr(f), N, Cn, f.shape[2], f.shape[3], ptr(rois[sel].contiguous()), sel.numel(),
                                              1.0 / s, PH, PW, ptr(o), stream_ptr()), "bd_roi_pool_max_fwd")
            out[sel] = o
        return out
    # ---- roi_align: one channel-last bf16 pyramid buffer, RoIs grouped per image into equal slot counts (label -1 = empty slot)
    Cp = (Cn + 7) // 8 * 8
    geom = ops.Geom(N, [f.shape[2] for f in features], [f.shape[3] for f in features])
    pyr = torch.zeros((N, geom.pix_per_img, Cp), dtype=torch.bfloat16, device=dev)
    for f, o, h, w in zip(features, geom.off, geom.H, geom.W):
        pyr[:, o:o + h * w, :Cn] = f.permute(0, 2, 3, 1).reshape(N, h * w, Cn).to(torch.bfloat16)
    bidx = rois[:, 0].long()
    counts = torch.bincount(bidx, minlength=N)
    rpi = max(int(counts.max().item()) if R else 0, 1)
    order = torch.argsort(bidx, stable=True)
    start = torch.cumsum(counts, 0) - counts
    slot = torch.empty((R,), dtype=torch.long, device=dev)
    slot[order] = bidx[order] * rpi + (torch.arange(R, device=dev) - start[bidx[order]])
    boxes = torch.zeros((N * rpi, 4), dtype=torch.float32, device=dev)
    labels = torch.full((N * rpi,), -1, dtype=torch.int32, device=dev)
    boxes[slot] = rois[:, 1:5]
    labels[slot] = 1
    out = torch.empty((N * rpi, PH * PW, Cp), dtype=torch.bfloat16, device=dev)
    # integer strides in the ABI: a fractional stride 1/k (tests/layers/test_roi_pool.py:71) is stride 1 on k-times larger boxes
    k = 1.0
    if min(strides) < 1:
        k = 1.0 / min(strides)
        boxes = boxes * k
    istr = [int(round(s * k)) for s in strides]
    ops.roi_align_fwd(pyr.reshape(N * geom.pix_per_img, Cp), geom, len(strides), istr, Cp, boxes, labels, rpi, (PH, PW), 2, out)
    return out[slot][:, :, :Cn].float().reshape(R, PH, PW, Cn).permute(0, 3, 1, 2).contiguous()


def sample_labels(labels, num_samples, label_value, ignore_label=-1, keys=None):
    """layers/common/sampling.py:7-30: keep at most `num_samples` of the entries equal to `label_value`, the others become
    `ignore_label`.  The reference draws megengine.random.uniform keys; here `keys` (fp32 in [0, 1), same shape) may be supplied for
    a reproducible choice -- default: torch.rand on the device.  The entries with the smallest keys survive (bd_sample_labels)."""
    assert labels.ndim == 1, "Only tensor of dim 1 is supported."
    mask = labels == label_value
    if int(mask.sum().item()) <= num_samples:
        return labels
    A = labels.shape[0]
    if keys is None:
        keys = torch.rand((A,), dtype=torch.float32, device=labels.device)
    tmp = torch.where(mask, 1, -1).to(torch.int32).reshape(1, A).contiguous()
    nv = torch.zeros((1,), dtype=torch.int32, device=labels.device)
    k = keys.float().reshape(1, A).contiguous()
    ops.sample_labels(tmp, k, k, int(num_samples), int(num_samples), nv)
    dropped = mask & (tmp[0] != 1)
    labels[dropped] = ignore_label
    return labels
