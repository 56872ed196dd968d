"""Thin torch-tensor wrappers over the C ABI (include/basedet_hip.h).  torch only carries memory and the stream.

Activations are "pixel-major" bf16 tensors of shape (N * pix_per_img, C) described by a `Geom`.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import _lib
from ._lib import ConvDesc, EPI_ADD_AFTER, EPI_ADD_BEFORE, EPI_MASK, EPI_RELU, EPI_SPARSE, check, f32arr, i32arr, ptr, stream_ptr


@dataclass
class Geom:
    """Pixel layout of an NHWC activation that may hold several pyramid levels per image."""
    N: int
    H: List[int]
    W: List[int]
    off: List[int] = field(default_factory=list)   # pixel offset of each level inside one image
    pix_per_img: int = 0

    def __post_init__(self):
        if not self.off:
            o = 0
            self.off = []
            for h, w in zip(self.H, self.W):
                self.off.append(o)
                o += h * w
            if not self.pix_per_img:
                self.pix_per_img = o
        if not self.pix_per_img:
            self.pix_per_img = sum(h * w for h, w in zip(self.H, self.W))

    @property
    def nlev(self):
        return len(self.H)

    @property
    def pixels(self):
        return self.N * self.pix_per_img

    def level(self, i):
        """Single-level view that keeps the parent's strides."""
        return Geom(self.N, [self.H[i]], [self.W[i]], [self.off[i]], self.pix_per_img)

    def conv_out(self, R, stride, pad):
        H = [(h + 2 * pad - R) // stride + 1 for h in self.H]
        W = [(w + 2 * pad - R) // stride + 1 for w in self.W]
        return Geom(self.N, H, W)


def single(N, H, W):
    return Geom(N, [H], [W])


def conv_desc(gin: Geom, gout: Geom, Cin, Cout, R, S, stride, pad) -> ConvDesc:
    d = ConvDesc()
    d.N, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad, d.nseg = gin.N, Cin, Cout, R, S, stride, pad, gin.nlev
    assert gin.nlev == gout.nlev <= _lib.BD_MAX_SEGS
    for i in range(gin.nlev):
        d.Hi[i], d.Wi[i], d.Ho[i], d.Wo[i] = gin.H[i], gin.W[i], gout.H[i], gout.W[i]
        d.in_off[i], d.out_off[i] = gin.off[i], gout.off[i]
    d.in_pix_per_img, d.out_pix_per_img = gin.pix_per_img, gout.pix_per_img
    return d


def L():
    return _lib.load()


# ---- per-call kernel routing (bd_conv_desc.route / .sr_seed) ---------------------------------------------------------------------------
# The library keeps no routing state (round 6: the bd_*_set_* knobs are gone).  A test or an A/B harness that wants a route sets it HERE;
# every wrapper below that takes a descriptor stamps the current words into it right before the call.  Production code never touches this.
_ROUTE = [0, 0, 0, 0]          # (route + 1) words: dense 1x1 mode, 3x3 / generic bit mask, weight-gradient bit mask, fp8 forward patch kernel
_SR_SEED = [0]                 # e5m2 stochastic-rounding seed of the calls made from now on (the fp8 model sets it per step)
_ROUTE_NAMES = {"dense1x1": 0, "patch3x3": 1, "wgrad": 2, "fp8_patch": 3}


def set_route(**kw):
    """set_route(dense1x1=mode, patch3x3=mask, wgrad=mask, fp8_patch=0|1); a value of None returns that word to the library's default.
    The meanings: include/basedet_hip.h, bd_conv_desc.route."""
    for k, v in kw.items():
        _ROUTE[_ROUTE_NAMES[k]] = 0 if v is None else int(v) + 1


def reset_route():
    _ROUTE[:] = [0, 0, 0, 0]
    _SR_SEED[0] = 0


def _stamp(d):
    d.route[0], d.route[1], d.route[2], d.route[3] = _ROUTE
    d.sr_seed = _SR_SEED[0]
    return d


# ---- dense path -------------------------------------------------------------------------------------------
def conv2d_fwd(d, x, w_packed, bias, y, add=None, flags=0, bits=None, y8=None, q_scale=1.0):
    """bits: optional uint32 [Cout/32][M] output, the bit-packed ReLU mask of y; y8: optional uint8 e4m3 twin of y (dense 1x1 launches
    only: bd_conv2d_fwd_bits / bd_conv2d_fwd_ex)."""
    if y8 is not None:
        check(L().bd_conv2d_fwd_ex(C.byref(_stamp(d)), ptr(x), ptr(w_packed), ptr(bias), ptr(add), ptr(y), ptr(bits), ptr(y8), float(q_scale), flags,
                                   stream_ptr()), "bd_conv2d_fwd_ex")
        return y
    if bits is not None:
        check(L().bd_conv2d_fwd_bits(C.byref(_stamp(d)), ptr(x), ptr(w_packed), ptr(bias), ptr(add), ptr(y), ptr(bits), flags, stream_ptr()),
              "bd_conv2d_fwd_bits")
        return y
    check(L().bd_conv2d_fwd(C.byref(_stamp(d)), ptr(x), ptr(w_packed), ptr(bias), ptr(add), ptr(y), flags, stream_ptr()), "bd_conv2d_fwd")
    return y


def conv2d_dgrad(d, g, w_packed_t, dx, add=None, mask=None, flags=0, maskbits=None, dx8=None, q_scale=1.0):
    """maskbits: the ReLU mask as written by conv2d_fwd(bits=...) instead of the bf16 activation `mask` (bd_conv2d_dgrad_bits);
    dx8: optional uint8 e5m2 twin of dx * q_scale (dense 1x1 launches only: bd_conv2d_dgrad_ex)."""
    if dx8 is not None:
        check(L().bd_conv2d_dgrad_ex(C.byref(_stamp(d)), ptr(g), ptr(w_packed_t), ptr(add), ptr(mask), ptr(maskbits), ptr(dx), ptr(dx8),
                                     float(q_scale), flags, stream_ptr()), "bd_conv2d_dgrad_ex")
        return dx
    if maskbits is not None:
        check(L().bd_conv2d_dgrad_bits(C.byref(_stamp(d)), ptr(g), ptr(w_packed_t), ptr(add), ptr(maskbits), ptr(dx), flags, stream_ptr()),
              "bd_conv2d_dgrad_bits")
        return dx
    check(L().bd_conv2d_dgrad(C.byref(_stamp(d)), ptr(g), ptr(w_packed_t), ptr(add), ptr(mask), ptr(dx), flags, stream_ptr()), "bd_conv2d_dgrad")
    return dx


def quantize_fp8(x, scale, q):
    """q (uint8, same element count) = e4m3(clamp(x * scale)) of a bf16 tensor."""
    check(L().bd_quantize_fp8(ptr(x), x.numel(), float(scale), ptr(q), stream_ptr()), "bd_quantize_fp8")
    return q


def weight_pack_fp8(w, row_scale, Cout, RS, Cin, act_scale, wq, wscale):
    check(L().bd_weight_pack_fp8(ptr(w), ptr(row_scale), Cout, RS, Cin, float(act_scale), ptr(wq), ptr(wscale), stream_ptr()),
          "bd_weight_pack_fp8")


def conv2d_fwd_fp8(d, xq, wq, wscale, bias, y, add=None, flags=0, y8=None, q_scale=1.0):
    """y8: optional uint8 twin of y (e4m3(y * q_scale)) for a following fp8 convolution."""
    check(L().bd_conv2d_fwd_fp8_ex(C.byref(_stamp(d)), ptr(xq), ptr(wq), ptr(wscale), ptr(bias), ptr(add), ptr(y), ptr(y8), float(q_scale), flags,
                                   stream_ptr()), "bd_conv2d_fwd_fp8")
    return y


def quantize_bf8(x, scale, q):
    """q (uint8) = e5m2(clamp(x * scale)) of a bf16 tensor (gradients); rounding: fp8_set_stochastic_rounding."""
    check(L().bd_quantize_bf8(ptr(x), x.numel(), float(scale), ptr(q), _SR_SEED[0], stream_ptr()), "bd_quantize_bf8")
    return q


def fp8_set_stochastic_rounding(seed):
    """seed != 0: the e5m2 quantisers CALLED from now on round stochastically (hash of seed and element index); 0: to nearest.  Python-side
    state: the seed travels with every call (bd_conv_desc.sr_seed, bd_quantize_bf8's argument)."""
    _SR_SEED[0] = int(seed) & 0xFFFFFFFF


def absmax_bf16(x, out):
    """out[0] (fp32, device) = max(out[0], max |x|) of a bf16 tensor."""
    check(L().bd_absmax_bf16(ptr(x), x.numel(), ptr(out), stream_ptr()), "bd_absmax_bf16")


def weight_pack_fp8_t(w, row_scale, Cout, RS, Cin, grad_scale, wq_t, wscale_t):
    check(L().bd_weight_pack_fp8_t(ptr(w), ptr(row_scale), Cout, RS, Cin, float(grad_scale), ptr(wq_t), ptr(wscale_t), stream_ptr()),
          "bd_weight_pack_fp8_t")


def conv2d_dgrad_fp8(d, g8, wq_t, wscale_t, dx, add=None, mask=None, flags=0, dx8=None, q_scale=1.0):
    check(L().bd_conv2d_dgrad_fp8(C.byref(_stamp(d)), ptr(g8), ptr(wq_t), ptr(wscale_t), ptr(add), ptr(mask), ptr(dx), ptr(dx8), float(q_scale),
                                  flags, stream_ptr()), "bd_conv2d_dgrad_fp8")
    return dx


def conv1x1_fp8(d, mode, xq, wq, wscale, bias, y, add=None, mask=None, maskbits=None, bits=None, y8=None, q_scale=1.0, flags=0):
    """Dense 1x1 launch on one-byte operands: mode 0 forward (xq e4m3), mode 1 data gradient (xq = e5m2 gradient)."""
    check(L().bd_conv1x1_fp8(C.byref(_stamp(d)), mode, ptr(xq), ptr(wq), ptr(wscale), ptr(bias), ptr(add), ptr(mask), ptr(maskbits), ptr(y),
                             ptr(bits), ptr(y8), float(q_scale), flags, stream_ptr()), "bd_conv1x1_fp8")
    return y


def conv2d_wgrad_fp8_workspace_bytes(d):
    return int(L().bd_conv2d_wgrad_fp8_workspace_bytes(C.byref(_stamp(d))))


def conv2d_wgrad_fp8(d, x8, g8, inv_scale, dw, ws, row_scale=None, accumulate=False):
    """Weight gradient of a 3x3 / stride-1 convolution from the one-byte twins (x8 e4m3, g8 e5m2); dw fp32 [Cout][9][Cin]."""
    check(L().bd_conv2d_wgrad_fp8(C.byref(_stamp(d)), ptr(x8), ptr(g8), float(inv_scale), ptr(row_scale), ptr(dw), int(accumulate), ptr(ws),
                                  ws.numel() * ws.element_size(), stream_ptr()), "bd_conv2d_wgrad_fp8")
    return dw


def conv1x1_fp8_ok(d, mode):
    """True when bd_conv1x1_fp8 takes this descriptor."""
    K, CO = (d.Cin, d.Cout) if mode == 0 else (d.Cout, d.Cin)
    return dense_1x1_bits_ok(d) and K % 128 == 0 and CO % 32 == 0


def fp8_dgrad_ok(d):
    """True when bd_conv2d_dgrad_fp8 takes this descriptor (conv3x3_pp8.hip, mode 1)."""
    same = all(d.Hi[i] == d.Ho[i] and d.Wi[i] == d.Wo[i] for i in range(d.nseg))
    return d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1 and same and d.Cin > 128 and d.Cin % 8 == 0 and d.Cout % 16 == 0


def dense_1x1_bits_ok(d):
    """True when bd_conv2d_fwd_bits / bd_conv2d_dgrad_bits take this descriptor (conv1x1.hip)."""
    return (d.R == 1 and d.S == 1 and d.stride == 1 and d.pad == 0 and d.nseg == 1 and d.in_off[0] == 0 and d.out_off[0] == 0
            and d.in_pix_per_img == d.Hi[0] * d.Wi[0] and d.out_pix_per_img == d.Ho[0] * d.Wo[0]
            and d.Cin % 32 == 0 and d.Cout % 32 == 0 and d.N * d.in_pix_per_img * max(d.Cin, d.Cout) * 2 < 0x7fffffff)


def conv2d_wgrad_workspace_bytes(d):
    return int(L().bd_conv2d_wgrad_workspace_bytes(C.byref(_stamp(d))))


def conv2d_wgrad(d, x, g, dw, ws, row_scale=None, accumulate=False):
    check(L().bd_conv2d_wgrad(C.byref(_stamp(d)), ptr(x), ptr(g), ptr(row_scale), ptr(dw), int(accumulate), ptr(ws),
                              ws.numel() * ws.element_size(), stream_ptr()), "bd_conv2d_wgrad")
    return dw


def conv2d_wgrad_bias_workspace_bytes(d):
    return int(L().bd_conv2d_wgrad_bias_workspace_bytes(C.byref(_stamp(d))))


def conv2d_wgrad_bias(d, x, g, dw, dbias, ws, row_scale=None, accumulate=False):
    """Weight and bias gradient in one call (the 3x3 patch kernel sums g's columns while staging them)."""
    check(L().bd_conv2d_wgrad_bias(C.byref(_stamp(d)), ptr(x), ptr(g), ptr(row_scale), ptr(dw), ptr(dbias), int(accumulate), ptr(ws),
                                   ws.numel() * ws.element_size(), stream_ptr()), "bd_conv2d_wgrad_bias")
    return dw


def stem_conv7x7_fwd(N, H, W, x_halo, w_stem, bias, y):
    check(L().bd_stem_conv7x7_fwd(N, H, W, ptr(x_halo), ptr(w_stem), ptr(bias), ptr(y), stream_ptr()), "bd_stem_conv7x7_fwd")
    return y


def stem_weight_pack(w, row_scale, w_stem):
    check(L().bd_stem_weight_pack(ptr(w), ptr(row_scale), ptr(w_stem), stream_ptr()), "bd_stem_weight_pack")


def weight_pack(w, row_scale, w_fwd, w_dgrad, Cout, RS, Cin):
    check(L().bd_weight_pack(ptr(w), ptr(row_scale), ptr(w_fwd), ptr(w_dgrad), Cout, RS, Cin, stream_ptr()), "bd_weight_pack")


def build_pack_table(entries, device):
    """entries: [(w, row_scale | None, w_fwd | None, w_dgrad | None, Cout, RS, Cin)] -> (device byte tensor of bd_pack_desc, n,
    total blocks).  The tensors must stay alive (and in place) as long as the table is used."""
    import ctypes as C
    import numpy as np
    from ._lib import PackDesc
    arr = (PackDesc * len(entries))()
    start = 0
    for i, (w, rs, wf, wd, co, r, ci) in enumerate(entries):
        arr[i].w = w.data_ptr()
        arr[i].row_scale = rs.data_ptr() if rs is not None else None
        arr[i].w_fwd = wf.data_ptr() if wf is not None else None
        arr[i].w_dgrad = wd.data_ptr() if wd is not None else None
        arr[i].Cout, arr[i].RS, arr[i].Cin, arr[i].block_start = co, r, ci, start
        start += int(L().bd_weight_pack_blocks(co, r, ci))
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device), len(entries), start


def weight_pack_multi(table):
    t, n, blocks = table
    check(L().bd_weight_pack_multi(ptr(t), n, blocks, stream_ptr()), "bd_weight_pack_multi")


def colsum_workspace_bytes(Cn):
    return int(L().bd_colsum_workspace_bytes(Cn))


def colsum_bf16(g, rows, Cn, out, ws, accumulate=False, geom=None):
    """geom: single-level Geom selecting the pixel rows of one pyramid level (default: all `rows` rows)."""
    if geom is None:
        n, ppi, off, cnt = 1, rows, 0, rows
    else:
        assert geom.nlev == 1
        n, ppi, off, cnt = geom.N, geom.pix_per_img, geom.off[0], geom.H[0] * geom.W[0]
    check(L().bd_colsum_bf16(ptr(g), n, ppi, off, cnt, Cn, ptr(out), int(accumulate), ptr(ws), ws.numel() * ws.element_size(),
                             stream_ptr()), "bd_colsum_bf16")
    return out


# ---- image ops --------------------------------------------------------------------------------------------
def pad_normalize(x, Hp, Wp, mean, std, out):
    N, _, H, W = x.shape
    check(L().bd_pad_normalize(ptr(x), N, H, W, Hp, Wp, f32arr(mean), f32arr(std), ptr(out), stream_ptr()), "bd_pad_normalize")
    return out


def pad_normalize_nchw(x, Hp, Wp, mean, std, out):
    N, _, H, W = x.shape
    check(L().bd_pad_normalize_nchw(ptr(x), N, H, W, Hp, Wp, f32arr(mean), f32arr(std), ptr(out), stream_ptr()), "bd_pad_normalize_nchw")
    return out


def stem_pool_fwd(N, H, W, x_halo, w_stem, bias, y_pool):
    """stem_conv7x7_fwd + maxpool3x3s2_fwd in one launch (bit-identical; the half-resolution tensor is never written)."""
    check(L().bd_stem_pool_fwd(N, H, W, ptr(x_halo), ptr(w_stem), ptr(bias), ptr(y_pool), stream_ptr()), "bd_stem_pool_fwd")


def maxpool3x3s2_fwd(x, N, H, W, Cn, y):
    check(L().bd_maxpool3x3s2_fwd(ptr(x), N, H, W, Cn, ptr(y), stream_ptr()), "bd_maxpool3x3s2_fwd")
    return y


def upsample2x_add_fwd(top, gtop: Geom, lat, glat: Geom, Cn):
    check(L().bd_upsample2x_add_fwd(ptr(top), gtop.pix_per_img, gtop.off[0], ptr(lat), glat.pix_per_img, glat.off[0],
                                    gtop.N, gtop.H[0], gtop.W[0], Cn, stream_ptr()), "bd_upsample2x_add_fwd")


def upsample2x_add_bwd(dlat, glat: Geom, dtop, gtop: Geom, Cn, accumulate):
    check(L().bd_upsample2x_add_bwd(ptr(dlat), glat.pix_per_img, glat.off[0], ptr(dtop), gtop.pix_per_img, gtop.off[0],
                                    gtop.N, gtop.H[0], gtop.W[0], Cn, int(accumulate), stream_ptr()), "bd_upsample2x_add_bwd")


def relu_bf16(x, y):
    check(L().bd_relu_bf16(ptr(x), ptr(y), x.numel(), stream_ptr()), "bd_relu_bf16")
    return y


def relu_bwd_bf16(g, mask, y, add=None):
    check(L().bd_relu_bwd_bf16(ptr(g), ptr(mask), ptr(add), ptr(y), g.numel(), stream_ptr()), "bd_relu_bwd_bf16")
    return y


def add_bf16(a, b, y):
    check(L().bd_add_bf16(ptr(a), ptr(b), ptr(y), a.numel(), stream_ptr()), "bd_add_bf16")
    return y


# ---- box ops ----------------------------------------------------------------------------------------------
def anchors_generate(H, W, stride, offset, base, out):
    check(L().bd_anchors_generate(H, W, stride, float(offset), ptr(base), base.shape[0], ptr(out), stream_ptr()), "bd_anchors_generate")
    return out


def points_generate(H, W, stride, offset, A, out):
    check(L().bd_points_generate(H, W, stride, float(offset), A, ptr(out), stream_ptr()), "bd_points_generate")
    return out


def box_pairwise(b1, b2, mode):
    m, n = b1.shape[0], b2.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=b1.device)
    check(L().bd_box_pairwise(ptr(b1), m, ptr(b2), n, mode, ptr(out), stream_ptr()), "bd_box_pairwise")
    return out


def box_encode(anchors, gt, mean, std):
    out = torch.empty_like(anchors)
    check(L().bd_box_encode(ptr(anchors), ptr(gt), anchors.shape[0], f32arr(mean), f32arr(std), ptr(out), stream_ptr()), "bd_box_encode")
    return out


def box_decode(anchors, deltas, mean, std):
    out = torch.empty_like(anchors)
    check(L().bd_box_decode(ptr(anchors), ptr(deltas), anchors.shape[0], f32arr(mean), f32arr(std), ptr(out), stream_ptr()), "bd_box_decode")
    return out


def retina_assign_encode(anchors, gt_boxes, num_gt, thr_lo, thr_hi, allow_lq, mean, std, labels, match_idx, offsets, num_fg, ws):
    A = anchors.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    check(L().bd_retina_assign_encode(ptr(anchors), A, ptr(gt_boxes), ptr(num_gt), N, Gmax, float(thr_lo), float(thr_hi),
                                      int(allow_lq), f32arr(mean), f32arr(std), ptr(labels), ptr(match_idx), ptr(offsets),
                                      ptr(num_fg), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "bd_retina_assign_encode")


def fcos_assign(points, lvl_start, soi, strides, radius, gt_boxes, num_gt, labels, offsets, ctrness, stats):
    P = points.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    Ln = len(strides)
    soi_flat = [float(v) for s in soi for v in s]
    check(L().bd_fcos_assign(ptr(points), P, i32arr(lvl_start), f32arr(soi_flat), i32arr(strides), Ln, float(radius),
                             ptr(gt_boxes), ptr(num_gt), N, Gmax, ptr(labels), ptr(offsets), ptr(ctrness), ptr(stats),
                             stream_ptr()), "bd_fcos_assign")


def atss_assign_workspace_bytes(N, P):
    return int(L().bd_atss_assign_workspace_bytes(N, P))


def atss_assign(points, lvl_start, strides, topk, anchor_scale, gt_boxes, num_gt, labels, offsets, ctrness, stats, ws):
    P = points.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    check(L().bd_atss_assign(ptr(points), P, i32arr(lvl_start), i32arr(strides), len(strides), int(topk), float(anchor_scale),
                             ptr(gt_boxes), ptr(num_gt), N, Gmax, ptr(labels), ptr(offsets), ptr(ctrness), ptr(stats), ptr(ws),
                             ws.numel() * ws.element_size(), stream_ptr()), "bd_atss_assign")


def ota_assign_workspace_bytes(N, P):
    return int(L().bd_ota_assign_workspace_bytes(N, P))


def ota_assign(points, lvl_start, strides, logits, K, pred_ltrb, gt_boxes, num_gt, alpha, gamma, reg_weight, center_radius,
               candidate_k, labels, targets, gt_ious, stats, ws):
    """OTA.get_ground_truth, top-k matcher (models/det/ota.py:76-181)."""
    P = points.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    check(L().bd_ota_assign(ptr(points), P, i32arr(lvl_start), i32arr(strides), len(strides), ptr(logits), int(K), ptr(pred_ltrb),
                            ptr(gt_boxes), ptr(num_gt), N, Gmax, float(alpha), float(gamma), float(reg_weight), float(center_radius),
                            int(candidate_k), ptr(labels), ptr(targets), ptr(gt_ious), ptr(stats), ptr(ws),
                            ws.numel() * ws.element_size(), stream_ptr()), "bd_ota_assign")


def ota_sinkhorn_workspace_bytes(N, P, Gmax):
    return int(L().bd_ota_sinkhorn_workspace_bytes(N, P, Gmax))


def ota_assign_sinkhorn(points, lvl_start, strides, logits, K, pred_ltrb, gt_boxes, num_gt, alpha, gamma, reg_weight, center_radius,
                        labels, targets, gt_ious, stats, ws, topq=20, eps=0.1, iters=50):
    """OTA.get_ground_truth with the SinkhornMatcher (models/det/ota.py:153-157, layers/common/matcher.py:106-121)."""
    P = points.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    check(L().bd_ota_assign_sinkhorn(ptr(points), P, i32arr(lvl_start), i32arr(strides), len(strides), ptr(logits), int(K), ptr(pred_ltrb),
                                     ptr(gt_boxes), ptr(num_gt), N, Gmax, float(alpha), float(gamma), float(reg_weight),
                                     float(center_radius), int(topq), float(eps), int(iters), ptr(labels), ptr(targets), ptr(gt_ious),
                                     ptr(stats), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "bd_ota_assign_sinkhorn")


def freeanchor_workspace_bytes(N, Gmax, bucket, A):
    return int(L().bd_freeanchor_workspace_bytes(N, Gmax, bucket, A))


def freeanchor_loss_fwd_bwd(logits, offsets, box_ld, anchors_per_pix, anchors, K, gt_boxes, num_gt, mean, std, iou_thresh, bucket,
                            beta, reg_weight, alpha, gamma, loss_out, d_logits, d_offsets, ws):
    """FreeAnchor bag losses + gradients (models/det/free_anchor.py:38-142); loss_out: fp32[2] = (pos_loss, neg_loss)."""
    A = anchors.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    check(L().bd_freeanchor_loss_fwd_bwd(ptr(logits), ptr(offsets), int(box_ld), int(anchors_per_pix), ptr(anchors), A, int(K),
                                         ptr(gt_boxes), ptr(num_gt), N, Gmax, f32arr(mean), f32arr(std), float(iou_thresh),
                                         int(bucket), float(beta), float(reg_weight), float(alpha), float(gamma), ptr(loss_out),
                                         ptr(d_logits), ptr(d_offsets), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()),
          "bd_freeanchor_loss_fwd_bwd")


def nms_workspace_bytes(n):
    return int(L().bd_nms_workspace_bytes(n))


def batched_nms(boxes, scores, idxs, iou_thresh, max_output=None):
    n = boxes.shape[0]
    dev = boxes.device
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    num = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((nms_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    check(L().bd_batched_nms(ptr(boxes), ptr(scores), ptr(idxs), n, float(iou_thresh), int(max_output or 0), ptr(keep),
                             ptr(num), ptr(ws), ws.numel(), stream_ptr()), "bd_batched_nms")
    return keep[: int(num.item())]


# ---- Faster R-CNN pieces ----------------------------------------------------------------------------------
def rpn_assign_encode(anchors, gt_boxes, num_gt, thr_lo, thr_hi, allow_lq, mean, std, labels, match_idx, offsets, num_fg, ws):
    A = anchors.shape[0]
    N, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
    check(L().bd_rpn_assign_encode(ptr(anchors), A, ptr(gt_boxes), ptr(num_gt), N, Gmax, float(thr_lo), float(thr_hi),
                                   int(allow_lq), f32arr(mean), f32arr(std), ptr(labels), ptr(match_idx), ptr(offsets),
                                   ptr(num_fg), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "bd_rpn_assign_encode")


def sample_labels(labels, keys_pos, keys_neg, num_pos_max, num_total, num_valid):
    N, A = labels.shape
    check(L().bd_sample_labels(ptr(labels), ptr(keys_pos), ptr(keys_neg), N, A, int(num_pos_max), int(num_total), ptr(num_valid),
                               stream_ptr()), "bd_sample_labels")


def segment_topk(scores, B, batch_stride, A, ldc, coff, seg_start, seg_rows, k, out_idx, out_score, out_cnt, min_score=None):
    check(L().bd_segment_topk(ptr(scores), int(scores.dtype == torch.bfloat16), B, batch_stride, A, ldc, coff, len(seg_start),
                              i32arr(seg_start), i32arr(seg_rows), k, float(min_score or 0.0), int(min_score is not None),
                              ptr(out_idx), ptr(out_score), ptr(out_cnt), stream_ptr()), "bd_segment_topk")


def nms_batched_workspace_bytes(B, Cn):
    return int(L().bd_nms_batched_workspace_bytes(B, Cn))


def nms_batched(boxes, scores, idxs, iou_thresh, max_output, keep, num_keep, ws):
    B, Cn = scores.shape
    check(L().bd_nms_batched(ptr(boxes), ptr(scores), ptr(idxs), B, Cn, float(iou_thresh), int(max_output or 0), keep.shape[1],
                             ptr(keep), ptr(num_keep), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "bd_nms_batched")


def rpn_proposals_workspace_bytes(N, lvl_pixels, A, pre_k, post_k):
    return int(L().bd_rpn_proposals_workspace_bytes(N, len(lvl_pixels), i32arr(lvl_pixels), A, pre_k, post_k))


def rpn_proposals(raw, ldc, A, cls_off, box_off, geom: Geom, anchors, im_info, mean, std, pre_k, nms_thresh, post_k, rois, num_rois, ws,
                  joint_nms=False):
    """joint_nms: the batched NMS as one problem per image (bd_rpn_proposals_joint: rounds 1-4) instead of level by level + merge."""
    lvl_pixels = [h * w for h, w in zip(geom.H, geom.W)]
    fn = L().bd_rpn_proposals_joint if joint_nms else L().bd_rpn_proposals
    check(fn(ptr(raw), ldc, A, cls_off, box_off, geom.N, geom.pix_per_img, geom.nlev, i32arr(geom.off),
             i32arr(lvl_pixels), ptr(anchors), ptr(im_info), im_info.shape[1], f32arr(mean), f32arr(std),
             int(pre_k), float(nms_thresh), int(post_k), ptr(rois), ptr(num_rois), ptr(ws),
             ws.numel() * ws.element_size(), stream_ptr()), "bd_rpn_proposals")


def rcnn_sample_targets(rois, num_rois, gt_boxes, num_gt, keys_fg, keys_bg, num_samples, num_fg_max, fg_thresh, bg_hi, bg_lo,
                        mean, std, out_rois, out_labels, out_targets, out_count, total_count):
    N, post_k = rois.shape[0], rois.shape[1]
    Gmax = gt_boxes.shape[1]
    check(L().bd_rcnn_sample_targets(ptr(rois), ptr(num_rois), post_k, ptr(gt_boxes), ptr(num_gt), N, Gmax, ptr(keys_fg),
                                     ptr(keys_bg), keys_fg.shape[1], int(num_samples), int(num_fg_max), float(fg_thresh),
                                     float(bg_hi), float(bg_lo), f32arr(mean), f32arr(std), ptr(out_rois), ptr(out_labels),
                                     ptr(out_targets), ptr(out_count), ptr(total_count), stream_ptr()), "bd_rcnn_sample_targets")


def conv1x1_thin_fwd(x, w, bias, M, Cin, Cout, y):
    check(L().bd_conv1x1_thin_fwd(ptr(x), ptr(w), ptr(bias), int(M), Cin, Cout, ptr(y), stream_ptr()), "bd_conv1x1_thin_fwd")


def conv1x1_thin_bwd_workspace_bytes():
    return int(L().bd_conv1x1_thin_bwd_workspace_bytes())


def conv1x1_thin_bwd(x, g, w, M, Cin, Cout, dx, dw, dbias, cout_real, ws):
    """Data + weight + bias gradient of a thin 1x1 prediction layer in one pass over its (ReLU-output) input: bd_conv1x1_thin_bwd."""
    check(L().bd_conv1x1_thin_bwd(ptr(x), ptr(g), ptr(w), int(M), Cin, Cout, ptr(dx), ptr(dw), ptr(dbias), cout_real, ptr(ws),
                                  ws.numel() * ws.element_size(), stream_ptr()), "bd_conv1x1_thin_bwd")


def roi_align_fwd(feat, geom: Geom, nlev, strides, Cn, rois, labels, rois_per_img, pool, sample_points, out):
    R = rois.shape[0]
    check(L().bd_roi_align_fwd(ptr(feat), geom.pix_per_img, Cn, nlev, i32arr(geom.off[:nlev]), i32arr(geom.H[:nlev]),
                               i32arr(geom.W[:nlev]), i32arr(strides[:nlev]), ptr(rois), ptr(labels), R, rois_per_img, pool[0],
                               pool[1], sample_points, ptr(out), stream_ptr()), "bd_roi_align_fwd")


def roi_align_bwd(gout, geom: Geom, nlev, strides, Cn, rois, labels, rois_per_img, pool, sample_points, gfeat):
    R = rois.shape[0]
    check(L().bd_roi_align_bwd(ptr(gout), geom.pix_per_img, Cn, nlev, i32arr(geom.off[:nlev]), i32arr(geom.H[:nlev]),
                               i32arr(geom.W[:nlev]), i32arr(strides[:nlev]), ptr(rois), ptr(labels), R, rois_per_img, pool[0],
                               pool[1], sample_points, ptr(gfeat), stream_ptr()), "bd_roi_align_bwd")


def roi_align_bwd_bf16_workspace_bytes(geom: Geom, rois_per_img):
    return int(L().bd_roi_align_bwd_bf16_workspace_bytes(geom.N, geom.nlev, i32arr(geom.H), i32arr(geom.W), rois_per_img))


def roi_align_bwd_bf16(gout, geom: Geom, nlev, strides, Cn, rois, labels, rois_per_img, pool, sample_points, gfeat, ws, accumulate=False):
    check(L().bd_roi_align_bwd_bf16(ptr(gout), geom.pix_per_img, Cn, nlev, geom.nlev, i32arr(geom.off), i32arr(geom.H), i32arr(geom.W),
                                    i32arr(strides[:nlev]), ptr(rois), ptr(labels), geom.N, rois_per_img, pool[0], pool[1], sample_points,
                                    ptr(gfeat), int(bool(accumulate)), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()),
          "bd_roi_align_bwd_bf16")


def subsample2x_fwd(src, gsrc: Geom, dst, gdst: Geom, Cn):
    check(L().bd_subsample2x_fwd(ptr(src), gsrc.pix_per_img, gsrc.off[0], gsrc.H[0], gsrc.W[0], ptr(dst), gdst.pix_per_img,
                                 gdst.off[0], Cn, gsrc.N, stream_ptr()), "bd_subsample2x_fwd")


def subsample2x_bwd_add(g_dst, gdst: Geom, g_src, gsrc: Geom, Cn):
    check(L().bd_subsample2x_bwd_add(ptr(g_dst), gdst.pix_per_img, gdst.off[0], ptr(g_src), gsrc.pix_per_img, gsrc.off[0],
                                     gsrc.H[0], gsrc.W[0], Cn, gsrc.N, stream_ptr()), "bd_subsample2x_bwd_add")


def f32_to_bf16(src, dst, accumulate=False):
    """dst = bf16(src), or bf16(float(dst) + src) with accumulate."""
    if accumulate:
        check(L().bd_f32_to_bf16_add(ptr(src), ptr(dst), src.numel(), stream_ptr()), "bd_f32_to_bf16_add")
    else:
        check(L().bd_f32_to_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "bd_f32_to_bf16")


def rpn_loss_fwd_bwd(raw, ldc, A, cls_off, box_off, labels, targets, rows, beta, num_valid, loss2, draw):
    check(L().bd_rpn_loss_fwd_bwd(ptr(raw), ldc, A, cls_off, box_off, ptr(labels), ptr(targets), rows, float(beta), ptr(num_valid),
                                  ptr(loss2), ptr(draw), stream_ptr()), "bd_rpn_loss_fwd_bwd")


def rcnn_loss_fwd_bwd(raw, ld, K, box_off, labels, targets, R, beta, num_samples, loss2, draw):
    check(L().bd_rcnn_loss_fwd_bwd(ptr(raw), ld, K, box_off, ptr(labels), ptr(targets), R, float(beta), ptr(num_samples),
                                   ptr(loss2), ptr(draw), stream_ptr()), "bd_rcnn_loss_fwd_bwd")


# ---- inference post-processing ----------------------------------------------------------------------------
def det_scores(logits, rows, K, scores, ctr=None, ctr_ld=1, ctr_off=0):
    check(L().bd_det_scores(ptr(logits), ptr(ctr), ctr_ld, ctr_off, rows, K, ptr(scores), stream_ptr()), "bd_det_scores")


def rcnn_predict(raw, ld, K, box_off, rois, num_rois, rois_per_img, mean, std, scores, boxes):
    R = raw.shape[0]
    check(L().bd_rcnn_predict(ptr(raw), ld, K, box_off, ptr(rois), ptr(num_rois), rois_per_img, R, f32arr(mean), f32arr(std),
                              ptr(scores), ptr(boxes), stream_ptr()), "bd_rcnn_predict")


def det_candidates(mode, topk_idx, topk_score, topk_cnt, Ln, k, lvl_row_off, K, anchors, offsets, off_ld, A, mean, std, item_boxes,
                   boxes, scores, labels):
    check(L().bd_det_candidates(mode, ptr(topk_idx), ptr(topk_score), ptr(topk_cnt), Ln, k, i32arr(lvl_row_off), K, ptr(anchors),
                                ptr(offsets), off_ld, A, f32arr(mean), f32arr(std), ptr(item_boxes), ptr(boxes), ptr(scores),
                                ptr(labels), stream_ptr()), "bd_det_candidates")


def det_finalize(boxes, scores, labels, keep, num_keep, max_out, im_info, out_boxes, out_scores, out_labels):
    check(L().bd_det_finalize(ptr(boxes), ptr(scores), ptr(labels), ptr(keep), ptr(num_keep), max_out, ptr(im_info), ptr(out_boxes),
                              ptr(out_scores), ptr(out_labels), stream_ptr()), "bd_det_finalize")


# ---- losses -----------------------------------------------------------------------------------------------
def focal_loss_fwd_bwd(logits, labels, rows, K, alpha, gamma, norm, grad_scale, loss_sum, dlogits, general=False):
    """general: the general-gamma kernel also for gamma == 2 (bd_focal_loss_fwd_bwd_general)"""
    fn = L().bd_focal_loss_fwd_bwd_general if general else L().bd_focal_loss_fwd_bwd
    check(fn(ptr(logits), ptr(labels), rows, K, float(alpha), float(gamma), ptr(norm),
             int(norm.dtype == torch.float32), float(grad_scale), ptr(loss_sum), ptr(dlogits), stream_ptr()), "bd_focal_loss_fwd_bwd")


def smooth_l1_fwd_bwd(pred, target, labels, pixels, A, ld, beta, norm, weight, loss_sum, dpred):
    check(L().bd_smooth_l1_fwd_bwd(ptr(pred), ptr(target), ptr(labels), pixels, A, ld, float(beta), ptr(norm),
                                   int(norm.dtype == torch.float32), float(weight), ptr(loss_sum), ptr(dpred), stream_ptr()),
          "bd_smooth_l1_fwd_bwd")


def giou_ltrb_fwd_bwd(pred, target, weight, labels, rows, norm, loss_weight, loss_sum, dpred):
    check(L().bd_giou_ltrb_fwd_bwd(ptr(pred), ptr(target), ptr(weight), ptr(labels), rows, ptr(norm), float(loss_weight),
                                   ptr(loss_sum), ptr(dpred), stream_ptr()), "bd_giou_ltrb_fwd_bwd")


def bce_logits_fwd_bwd(pred, target, labels, rows, norm, loss_sum, dpred, ld=1, off=0):
    check(L().bd_bce_logits_fwd_bwd(ptr(pred), ld, off, ptr(target), ptr(labels), rows, ptr(norm), ptr(loss_sum), ptr(dpred),
                                    stream_ptr()), "bd_bce_logits_fwd_bwd")


# ---- FCOS head pieces -------------------------------------------------------------------------------------
def groupnorm_workspace_bytes(N, Ln, Cn, pix_per_img):
    return int(L().bd_groupnorm_workspace_bytes(N, Ln, Cn, int(pix_per_img)))


def _lvl_arrays(geom: Geom):
    return i32arr(geom.off), i32arr([h * w for h, w in zip(geom.H, geom.W)])


def groupnorm_fwd(y, gamma, beta, geom: Geom, Cn, eps, relu, stats, z, ws):
    off, cnt = _lvl_arrays(geom)
    check(L().bd_groupnorm_fwd(ptr(y), ptr(gamma), ptr(beta), geom.N, geom.nlev, off, cnt, geom.pix_per_img, Cn, float(eps),
                               int(relu), ptr(stats), ptr(z), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()), "bd_groupnorm_fwd")
    return z


def conv2d_fwd_gnstats_bytes(d):
    return int(L().bd_conv2d_fwd_gnstats_bytes(C.byref(d)))


def conv2d_fwd_gnstats(d, x, w_packed, bias, y, part):
    """3x3 / stride-1 convolution into 256 channels that also leaves GroupNorm's per-patch statistics in `part` (fp32, conv2d_fwd_gnstats_bytes)."""
    check(L().bd_conv2d_fwd_gnstats(C.byref(_stamp(d)), ptr(x), ptr(w_packed), ptr(bias), ptr(y), ptr(part), part.numel() * part.element_size(),
                                    stream_ptr()), "bd_conv2d_fwd_gnstats")
    return y


def groupnorm_fwd_parts(d, y, part, gamma, beta, eps, relu, stats, z):
    """GroupNorm(32, 256) (+ReLU) from the statistics conv2d_fwd_gnstats left: stats (mean, rstd) [N][L][32][2] out, z out."""
    check(L().bd_groupnorm_fwd_parts(C.byref(d), ptr(y), ptr(part), ptr(gamma), ptr(beta), float(eps), int(relu), ptr(stats), ptr(z), stream_ptr()),
          "bd_groupnorm_fwd_parts")
    return z


def groupnorm_bwd(dz, y, gamma, beta, stats, geom: Geom, Cn, relu, dy, dgamma, dbeta, ws, accumulate=False):
    """The ReLU gate (relu=True) is recomputed from y, stats, gamma, beta: the forward's output z is not an operand."""
    off, cnt = _lvl_arrays(geom)
    check(L().bd_groupnorm_bwd(ptr(dz), ptr(y), ptr(gamma), ptr(beta), ptr(stats), geom.N, geom.nlev, off, cnt, geom.pix_per_img, Cn,
                               int(relu), ptr(dy), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), ws.numel() * ws.element_size(),
                               stream_ptr()), "bd_groupnorm_bwd")
    return dy


def fcos_offsets_fwd(raw, ld, scales, geom: Geom, strides, out):
    off, cnt = _lvl_arrays(geom)
    check(L().bd_fcos_offsets_fwd(ptr(raw), ld, ptr(scales), geom.N, geom.nlev, off, cnt, i32arr(strides), geom.pix_per_img, ptr(out),
                                  stream_ptr()), "bd_fcos_offsets_fwd")
    return out


def fcos_offsets_workspace_bytes():
    return int(L().bd_fcos_offsets_workspace_bytes())


def fcos_offsets_bwd(raw, ld, scales, geom: Geom, strides, d_off, d_ctr, d_raw, dscale, ws):
    off, cnt = _lvl_arrays(geom)
    check(L().bd_fcos_offsets_bwd(ptr(raw), ld, ptr(scales), geom.N, geom.nlev, off, cnt, i32arr(strides), geom.pix_per_img, ptr(d_off),
                                  ptr(d_ctr), ptr(d_raw), ptr(dscale), ptr(ws), ws.numel() * ws.element_size(), stream_ptr()),
          "bd_fcos_offsets_bwd")


def sgd_momentum_step(w, v, g, lr, momentum, wd, grad_scale=1.0):
    check(L().bd_sgd_momentum_step(ptr(w), ptr(v), ptr(g), w.numel(), float(lr), float(momentum), float(wd),
                                   float(grad_scale), stream_ptr()), "bd_sgd_momentum_step")


class WgradQueue:
    """bd_wgrad_queue_*: weight-gradient launches whose reduces are deferred to ONE launch per flush (the solver's gradient buckets).
    Every queued layer must keep its own workspace untouched until `flush`."""

    def __init__(self):
        self._h = C.c_void_p()
        check(L().bd_wgrad_queue_create(C.byref(self._h)), "bd_wgrad_queue_create")

    def wgrad(self, d, x, g, dw, dbias, ws, row_scale=None, accumulate=False):
        check(L().bd_conv2d_wgrad_queued(self._h, C.byref(_stamp(d)), ptr(x), ptr(g), ptr(row_scale), ptr(dw), ptr(dbias), int(accumulate), ptr(ws),
                                         ws.numel() * ws.element_size(), stream_ptr()), "bd_conv2d_wgrad_queued")

    def pending(self):
        return int(L().bd_wgrad_queue_pending(self._h))

    def flush(self):
        check(L().bd_wgrad_queue_flush(self._h, stream_ptr()), "bd_wgrad_queue_flush")

    def close(self):
        if self._h:
            L().bd_wgrad_queue_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:            # noqa: BLE001 (interpreter shutdown)
            pass


def clip_grad_value(g, lower, upper, pre_scale=1.0):
    """megengine.optimizer.clip_grad_value over the flat gradient arena, in place (engine/trainer.py:57-61)."""
    check(L().bd_clip_grad_value(ptr(g), g.numel(), float(pre_scale), float(lower), float(upper), stream_ptr()), "bd_clip_grad_value")


def clip_grad_norm_workspace_bytes():
    return int(L().bd_clip_grad_norm_workspace_bytes())


def clip_grad_norm(g, max_norm, ord=2.0, pre_scale=1.0, norm_out=None, ws=None):
    """megengine.optimizer.clip_grad_norm over the flat gradient arena, in place; the norm (of pre_scale * g) goes to norm_out[0]."""
    if ws is None:
        ws = torch.empty((clip_grad_norm_workspace_bytes(),), dtype=torch.uint8, device=g.device)
    check(L().bd_clip_grad_norm(ptr(g), g.numel(), float(pre_scale), float(max_norm), float(ord), ptr(norm_out), ptr(ws), ws.numel(),
                                stream_ptr()), "bd_clip_grad_norm")
    return norm_out


class HostStager:
    """bd_h2d_*: host batch (numpy float64 / float32 / uint8, C-contiguous) -> fp32 device tensor through the library's pinned staging
    buffer, converted by its worker threads and sent chunk by chunk (data_to_input's `Tensor(image)`, pre_processing.py:13)."""

    _DT = {"float64": 0, "float32": 1, "uint8": 2}

    def __init__(self, device, threads=0):
        self._h = C.c_void_p()
        self.device = torch.device(device)
        # torch.device("cuda") carries no index: the rank's CURRENT device is meant (one process per GPU), never device 0
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        check(L().bd_h2d_create(C.byref(self._h), int(index), int(threads)), "bd_h2d_create")
        self.threads = int(L().bd_h2d_threads(self._h))

    def supports(self, arr):
        return hasattr(arr, "dtype") and str(arr.dtype) in self._DT and getattr(arr, "flags", None) is not None and arr.flags["C_CONTIGUOUS"]

    def submit(self, arr, out=None, chunk_elems=0):
        """arr: numpy array; returns the fp32 device tensor of the same shape (valid in the current stream's order)."""
        if out is None:
            out = torch.empty(arr.shape, dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.numel() == arr.size and out.dtype == torch.float32
        check(L().bd_h2d_submit(self._h, C.c_void_p(arr.ctypes.data), self._DT[str(arr.dtype)], arr.size, ptr(out), int(chunk_elems),
                                stream_ptr()), "bd_h2d_submit")
        return out

    def close(self):
        if self._h:
            L().bd_h2d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:            # noqa: BLE001 (interpreter shutdown)
            pass


def bottleneck_fwd_supported(N, H, W, cin, cmid, cout, has_ds):
    return bool(L().bd_bottleneck_fwd_supported(int(N), int(H), int(W), int(cin), int(cmid), int(cout), int(bool(has_ds))))


def bottleneck_fwd(N, H, W, cin, cmid, cout, x, w1, b1, w2, b2, w3, b3, wd, bd, y):
    """A frozen Bottleneck block (models/cls/resnet.py:70-113) in one launch: bd_bottleneck_fwd."""
    check(L().bd_bottleneck_fwd(int(N), int(H), int(W), int(cin), int(cmid), int(cout), ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2),
                                ptr(w3), ptr(b3), ptr(wd), ptr(bd), ptr(y), stream_ptr()), "bd_bottleneck_fwd")
    return y
