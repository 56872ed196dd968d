"""Helpers shared by the GPU parity tests (layout conversions, reference convs on CPU fp32)."""
import numpy as np
import torch
import torch.nn.functional as TF


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nchw_to_pm(x):
    """(N,C,H,W) fp32 -> pixel-major (N*H*W, C) bf16 on cuda."""
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous().to(torch.bfloat16).cuda()


def pm_to_nchw(t, n, h, w):
    c = t.shape[1]
    return t.float().cpu().reshape(n, h, w, c).permute(0, 3, 1, 2).contiguous()


def oihw_to_ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous()


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def pack_weights(ops, w_oihw, row_scale=None):
    """fp32 OIHW (cpu) -> (w_fwd, w_dgrad) bf16 packed device tensors via bd_weight_pack."""
    co, ci, r, s = w_oihw.shape
    w = oihw_to_ohwi(w_oihw).cuda()
    wf = torch.empty((co, r * s, ci), dtype=torch.bfloat16, device="cuda")
    wd = torch.empty((ci, r * s, co), dtype=torch.bfloat16, device="cuda")
    rs = None if row_scale is None else row_scale.cuda()
    ops.weight_pack(w, rs, wf, wd, co, r * s, ci)
    return wf, wd
