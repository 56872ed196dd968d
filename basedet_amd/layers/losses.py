"""basedet.layers.losses on the HIP kernels -- same names, arguments, defaults and asserts as the reference
(layers/losses/{sigmoid_focal_loss,smooth_l1_loss,iou_loss,cross_entropy}.py).  Each function returns the ELEMENTWISE loss, as the
reference does (the caller reduces), and is differentiable w.r.t. the prediction: the backward of every function is the HIP kernel's
analytic gradient (a torch.autograd.Function carries it; torch computes nothing here).  The training step of the models does not go
through these -- it uses the fused label-driven kernels (bd_focal_loss_fwd_bwd, ...) -- they are the drop-in operator surface."""
import ctypes as C

import torch

from .. import ops
from .._lib import check, ptr, stream_ptr

_IOU_TYPES = {"iou": 0, "linear_iou": 1, "giou": 2, "square_iou": 3}


def _f32(t):
    return t.float().contiguous()


class _Elem(torch.autograd.Function):
    """loss = kernel(pred, target); d loss / d pred from the same kernel."""

    @staticmethod
    def forward(ctx, kind, pred, target, a, b):
        p, t = _f32(pred), _f32(target.expand_as(pred) if target.shape != pred.shape else target)
        out = torch.empty_like(p)
        _launch(kind, p, t, a, b, None, out, None)
        ctx.save_for_backward(p, t)
        ctx.kind, ctx.a, ctx.b, ctx.dtype = kind, a, b, pred.dtype
        return out

    @staticmethod
    def backward(ctx, gout):
        p, t = ctx.saved_tensors
        d = torch.empty_like(p)
        _launch(ctx.kind, p, t, ctx.a, ctx.b, _f32(gout), None, d)
        return None, d.to(ctx.dtype), None, None, None


def _launch(kind, p, t, a, b, gout, loss, d):
    L = ops.L()
    n = p.numel()
    if kind == "focal":
        check(L.bd_sigmoid_focal_loss_elem(ptr(p), ptr(t), n, float(a), float(b), ptr(gout), ptr(loss), ptr(d), stream_ptr()), kind)
    elif kind == "bce":
        check(L.bd_bce_elem(ptr(p), ptr(t), n, int(a), ptr(gout), ptr(loss), ptr(d), stream_ptr()), kind)
    else:
        check(L.bd_smooth_l1_elem(ptr(p), ptr(t), n, float(a), ptr(gout), ptr(loss), ptr(d), stream_ptr()), kind)


def sigmoid_focal_loss(logits, targets, alpha: float = -1, gamma: float = 0):
    """layers/losses/sigmoid_focal_loss.py:9-36."""
    return _Elem.apply("focal", logits, targets, alpha, gamma)


def binary_cross_entropy(pred, label, with_logits: bool = True):
    """layers/losses/cross_entropy.py:7-29."""
    return _Elem.apply("bce", pred, label, with_logits, 0)


def smooth_l1_loss(pred, target, beta: float = 1.0):
    """layers/losses/smooth_l1_loss.py:7-34."""
    return _Elem.apply("l1", pred, target, beta, 0)


class _IouLtrb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, loss_type, eps):
        p, t = _f32(pred).reshape(-1, 4), _f32(target).reshape(-1, 4)
        n = p.shape[0]
        loss = torch.empty(pred.shape[:-1], dtype=torch.float32, device=p.device)
        ious = torch.empty_like(loss)
        check(ops.L().bd_iou_loss_ltrb(ptr(p), ptr(t), n, loss_type, float(eps), None, ptr(loss), ptr(ious), None, stream_ptr()),
              "bd_iou_loss_ltrb")
        ctx.save_for_backward(p, t)
        ctx.args = (loss_type, eps, pred.shape, pred.dtype)
        ctx.mark_non_differentiable(ious)
        return loss, ious

    @staticmethod
    def backward(ctx, gloss, _gious):
        p, t = ctx.saved_tensors
        loss_type, eps, shape, dtype = ctx.args
        d = torch.empty_like(p)
        g = _f32(gloss).reshape(-1)
        check(ops.L().bd_iou_loss_ltrb(ptr(p), ptr(t), p.shape[0], loss_type, float(eps), ptr(g), None, None, ptr(d), stream_ptr()),
              "bd_iou_loss_ltrb")
        return d.reshape(shape).to(dtype), None, None, None


def _to_xyxy(boxes, box_mode):
    """BoxConverter.convert(boxes, mode + "2xyxy") (structures/box_convert.py) for the modes iou_loss accepts."""
    if box_mode == "xyxy":
        return boxes
    b = boxes
    if box_mode == "xywh":
        return torch.cat([b[..., :2], b[..., :2] + b[..., 2:]], dim=-1)
    if box_mode == "xcycwh":
        return torch.cat([b[..., :2] - b[..., 2:] / 2, b[..., :2] + b[..., 2:] / 2], dim=-1)
    raise AssertionError(f"{box_mode} not supported.")


def iou_loss(pred, target, box_mode: str = "xyxy", loss_type: str = "iou", eps: float = 1e-8, return_iou: bool = False):
    """layers/losses/iou_loss.py:59-105.  box_mode "ltrb": row-wise loss of (..., 4) distance boxes (differentiable w.r.t. pred);
    any other mode converts to xyxy and -- exactly as the reference -- evaluates Boxes(pred).iou / .giou(target), i.e. the PAIRWISE
    (N, M) matrix (the form HungarianMatcher consumes, layers/common/matcher.py:84), mapped through the loss; that branch is a
    forward-only cost (no gradient)."""
    assert loss_type in ["iou", "linear_iou", "giou", "square_iou"]
    lt = _IOU_TYPES[loss_type]
    if box_mode == "ltrb":
        loss, ious = _IouLtrb.apply(pred, target, lt, eps)
    else:
        assert box_mode in ("xyxy", "xywh", "xcycwh"), f"{box_mode} not supported."
        p = _f32(_to_xyxy(pred.detach(), box_mode)).reshape(-1, 4)
        t = _f32(_to_xyxy(target.detach(), box_mode)).reshape(-1, 4)
        ious = ops.box_pairwise(p, t, 3 if loss_type == "giou" else 0)
        loss = torch.empty_like(ious)
        check(ops.L().bd_iou_to_loss(ptr(ious), ious.numel(), lt, float(eps), ptr(loss), stream_ptr()), "bd_iou_to_loss")
    if return_iou:
        return loss, ious
    return loss


def weighted_cross_entropy(input, target, weight=None):  # noqa: A002 - reference argument name
    """layers/losses/cross_entropy.py:32-40: softmax cross entropy with per-class weights normalised by their mean over the batch,
    averaged.  Runs on bd_rcnn_loss_fwd_bwd's softmax-CE kernel per row when unweighted; the weighted form is a short composition of
    device tensor ops on top of the row losses (kept differentiable through torch for this small (N, K) head-side use)."""
    logz = torch.logsumexp(input.float(), dim=1)
    primary = input.float().gather(1, target.reshape(-1, 1).long()).squeeze(1)
    ce = logz - primary
    if weight is not None:
        w = weight[target.flatten().long()].reshape(target.shape).float()
        ce = ce * (w / w.mean())
    return ce.mean()
