"""Host-side runtime for the HIP training path: parameter arenas, packed-weight maintenance and the
conv layer object whose forward / dgrad / wgrad map 1:1 onto the C ABI (include/basedet_hip.h).

torch tensors are used as the memory carrier only -- no torch operator computes anything on this path
(allocation, zero-fill and views aside).
"""
import numpy as np
import torch

from .. import ops
from ..ops import Geom

BN_EPS = 1e-5  # basecore FrozenBatchNorm eps (un-vendored; documented choice, same as oracle/model.py)


class ParamArena:
    """Flat fp32 arenas (weights / gradients / momentum) for all trainable parameters: one SGD launch,
    contiguous buckets for the RCCL all-reduce (solver/default_solver.py:118-124)."""

    def __init__(self, device):
        self.device = device
        self.entries = []   # (name, shape, offset, numel)
        self.total = 0
        self.w = self.g = self.v = None

    def reserve(self, name, shape):
        n = int(np.prod(shape))
        off = self.total
        self.entries.append((name, tuple(shape), off, n))
        self.total += (n + 63) // 64 * 64     # 256-byte aligned slices
        return len(self.entries) - 1

    def allocate(self):
        self.w = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.g = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.v = torch.zeros(self.total, dtype=torch.float32, device=self.device)

    def view(self, which, idx):
        _, shape, off, n = self.entries[idx]
        return getattr(self, which)[off:off + n].view(shape)


class ConvLayer:
    """One convolution (+ folded FrozenBN or bias).  Master weight fp32 [Cout][R][S][Cin] (OHWI);
    bf16 packed copies for the forward ([Cout][RS][Cin], BN scale folded in) and dgrad ([Cin][RS][Cout]) kernels."""

    def __init__(self, name, cin, cout, k, stride, pad, device, bn_prefix=None, has_bias=False, trainable=True,
                 cout_pad=None):
        self.name, self.cin, self.cout_real, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.cout = cout_pad or cout          # padded channel count seen by the kernels (bbox_pred: 36 -> 40)
        self.bn_prefix, self.has_bias, self.trainable = bn_prefix, has_bias, trainable
        self.device = device
        self.w = self.b = self.gw = self.gb = None
        self.row_scale = None
        self.w_fwd = torch.empty((self.cout, k * k, cin), dtype=torch.bfloat16, device=device)
        self.w_dgrad = torch.empty((cin, k * k, self.cout), dtype=torch.bfloat16, device=device) if trainable else None
        self._desc_cache = {}
        # fp8 forward (BASELINE config 5, enable_fp8): e4m3 weights with one scale per output channel + the input cast to e4m3
        self.fp8 = False
        self.fp8_dgrad = False          # data gradient on the fp8 patch kernel too (e5m2 gradients under a static gradient scale)
        self.fp8_wgrad = False          # 3x3 weight gradient from the one-byte twins of x and g (bd_conv2d_wgrad_fp8) when both are handed over
        self.fp8_1x1 = False            # dense 1x1 launches on one-byte operands when the producer wrote the input's twin (bd_conv1x1_fp8)
        self.fp8_1x1_dgrad = False
        self.w_q8 = self.w_scale8 = self.w_q8t = self.w_scale8t = None
        self.act_scale = 1.0
        self.grad_scale = 4096.0
        self.amax_slot = None          # delayed scaling of the e5m2 gradients (fpn_base): where max |g| goes on probe steps
        self.probe_ctl = None
        self.q8_scratch = None          # callable(nbytes) -> uint8 scratch tensor on the stream this layer's forward runs on

    # -- parameters -------------------------------------------------------------------------------
    def bind(self, arena, params):
        """Load numpy parameters (reference layout) into arena views / standalone tensors."""
        w = torch.from_numpy(params[self.name + ".weight"]).permute(0, 2, 3, 1).contiguous()   # OIHW -> OHWI
        if self.cout != self.cout_real:
            pad = torch.zeros((self.cout - self.cout_real,) + tuple(w.shape[1:]))
            w = torch.cat([w, pad], 0)
        if self.trainable:
            self.w = arena.view("w", self._wi); self.gw = arena.view("g", self._wi)
            self.w.copy_(w)
        else:
            self.w = w.to(self.device)
        if self.bn_prefix:
            g, b = params[self.bn_prefix + ".weight"], params[self.bn_prefix + ".bias"]
            m, v = params[self.bn_prefix + ".running_mean"], params[self.bn_prefix + ".running_var"]
            scale = g / np.sqrt(v + BN_EPS)
            self.row_scale = torch.from_numpy(scale.astype(np.float32)).to(self.device)
            self.b = torch.from_numpy((b - m * scale).astype(np.float32)).to(self.device)
        elif self.has_bias:
            b = torch.from_numpy(params[self.name + ".bias"])
            if self.cout != self.cout_real:
                b = torch.cat([b, torch.zeros(self.cout - self.cout_real)])
            if self.trainable:
                self.b = arena.view("w", self._bi); self.gb = arena.view("g", self._bi)
                self.b.copy_(b)
            else:
                self.b = b.to(self.device)

    def reserve(self, arena):
        if not self.trainable:
            return
        self._wi = arena.reserve(self.name + ".weight", (self.cout, self.k, self.k, self.cin))
        if self.has_bias:
            self._bi = arena.reserve(self.name + ".bias", (self.cout,))

    def export(self, out):
        """Back to the reference layout (OIHW numpy)."""
        out[self.name + ".weight"] = self.w[: self.cout_real].permute(0, 3, 1, 2).contiguous().cpu().numpy()
        if self.has_bias and not self.bn_prefix:
            out[self.name + ".bias"] = self.b[: self.cout_real].cpu().numpy().copy()

    def export_grad(self, out):
        out[self.name + ".weight"] = self.gw[: self.cout_real].permute(0, 3, 1, 2).contiguous().cpu()
        if self.gb is not None:
            out[self.name + ".bias"] = self.gb[: self.cout_real].cpu().clone()

    def enable_fp8(self, q8_scratch, act_scale=1.0, dgrad=False, grad_scale=4096.0, wgrad=False):
        """Forward through bd_conv2d_fwd_fp8 (csrc/conv3x3_pp8.hip / conv_fp8.hip); with dgrad=True the data gradient through
        bd_conv2d_dgrad_fp8 as well (3x3 / stride 1 / Cin > 128).  The weight gradient keeps the bf16 activations and gradients."""
        assert self.cin % 16 == 0 and self.cout % 8 == 0
        self.fp8, self.q8_scratch, self.act_scale = True, q8_scratch, float(act_scale)
        self.w_q8 = torch.empty((self.cout, self.k * self.k, self.cin), dtype=torch.uint8, device=self.device)
        self.w_scale8 = torch.empty((self.cout,), dtype=torch.float32, device=self.device)
        if wgrad and self.trainable and self.k == 3 and self.stride == 1 and self.cin % 16 == 0 and self.cout % 16 == 0:
            self.fp8_wgrad, self.grad_scale = True, float(grad_scale)
        if dgrad and self.trainable and self.k == 3 and self.stride == 1 and self.cin > 128 and self.cout % 16 == 0:
            self.fp8_dgrad, self.grad_scale = True, float(grad_scale)
            self.w_q8t = torch.empty((self.cin, self.k * self.k, self.cout), dtype=torch.uint8, device=self.device)
            self.w_scale8t = torch.empty((self.cin,), dtype=torch.float32, device=self.device)

    def enable_fp8_1x1(self, act_scale=1.0, dgrad=False, grad_scale=4096.0, expanding=True):
        """1x1 / stride 1 layers: forward through bd_conv1x1_fp8 whenever the caller hands over the e4m3 twin of the input (never a cast
        pass: the launch is memory-bound, a cast would cost what the one-byte operand saves), with dgrad=True the data gradient too
        (from the e5m2 twin of the output gradient).  Weight gradients stay bf16."""
        assert self.k == 1 and self.stride == 1 and self.cin % 32 == 0 and self.cout % 32 == 0
        self.act_scale, self.grad_scale = float(act_scale), float(grad_scale)
        # expanding=False keeps only the REDUCING direction (K >= 2 x produced channels: conv1 forward, conv3's data gradient), where the
        # one-byte operand is most of the launch's bytes; the expanding direction is all epilogue (bf16 residual in, bf16 + twin + gate
        # bits out).  See FPNDetector._build_layers for what was measured
        if self.cin % 128 == 0 and (self.cin >= 2 * self.cout or expanding):
            self.fp8_1x1 = True
            self.w_q8 = torch.empty((self.cout, 1, self.cin), dtype=torch.uint8, device=self.device)
            self.w_scale8 = torch.empty((self.cout,), dtype=torch.float32, device=self.device)
        if dgrad and self.trainable and self.cout % 128 == 0 and (self.cout >= 2 * self.cin or expanding):
            self.fp8_1x1_dgrad = True
            self.w_q8t = torch.empty((self.cin, 1, self.cout), dtype=torch.uint8, device=self.device)
            self.w_scale8t = torch.empty((self.cin,), dtype=torch.float32, device=self.device)

    def pack_fp8(self):
        if self.k == 1 and (self.fp8_1x1 or self.fp8_1x1_dgrad):
            if self.fp8_1x1:
                ops.weight_pack_fp8(self.w, self.row_scale, self.cout, 1, self.cin, self.act_scale, self.w_q8, self.w_scale8)
            if self.fp8_1x1_dgrad:
                ops.weight_pack_fp8_t(self.w, self.row_scale, self.cout, 1, self.cin, self.grad_scale, self.w_q8t, self.w_scale8t)
            return
        ops.weight_pack_fp8(self.w, self.row_scale, self.cout, self.k * self.k, self.cin, self.act_scale, self.w_q8, self.w_scale8)
        if self.fp8_dgrad:
            ops.weight_pack_fp8_t(self.w, self.row_scale, self.cout, self.k * self.k, self.cin, self.grad_scale, self.w_q8t, self.w_scale8t)

    def pack(self):
        ops.weight_pack(self.w, self.row_scale, self.w_fwd, self.w_dgrad, self.cout, self.k * self.k, self.cin)
        if self.fp8 or self.fp8_1x1 or self.fp8_1x1_dgrad:
            self.pack_fp8()

    # -- kernels ----------------------------------------------------------------------------------
    def desc(self, gin: Geom, gout: Geom):
        key = (gin.N, tuple(gin.H), tuple(gin.W), tuple(gin.off), gin.pix_per_img,
               tuple(gout.H), tuple(gout.W), tuple(gout.off), gout.pix_per_img)
        d = self._desc_cache.get(key)
        if d is None:
            d = ops.conv_desc(gin, gout, self.cin, self.cout, self.k, self.k, self.stride, self.pad)
            self._desc_cache[key] = d
        return d

    def forward(self, x, gin, gout, y, add=None, relu=False, bits=None, x8=None, y8=None, q_scale=1.0):
        """x8: the e4m3 twin of x when a producing fp8 launch wrote one (else x is cast by bd_quantize_fp8); y8: twin of y to write
        for a following fp8 convolution.  Both are ignored on the bf16 path."""
        flags = (ops.EPI_RELU if relu else 0) | (ops.EPI_ADD_BEFORE if add is not None else 0)
        if self.fp8_1x1 and x8 is not None and ops.conv1x1_fp8_ok(self.desc(gin, gout), 0):
            return ops.conv1x1_fp8(self.desc(gin, gout), 0, x8, self.w_q8, self.w_scale8, self.b, y, add=add, bits=bits, y8=y8,
                                   q_scale=q_scale, flags=flags)
        if self.fp8 and bits is None:
            xq = x8 if x8 is not None else ops.quantize_fp8(x, self.act_scale, self.q8_scratch(x.numel())[: x.numel()])
            return ops.conv2d_fwd_fp8(self.desc(gin, gout), xq, self.w_q8, self.w_scale8, self.b, y, add=add, flags=flags, y8=y8,
                                      q_scale=self.act_scale)
        return ops.conv2d_fwd(self.desc(gin, gout), x, self.w_fwd, self.b, y, add=add, flags=flags, bits=bits,
                              y8=y8 if (y8 is not None and ops.dense_1x1_bits_ok(self.desc(gin, gout))) else None, q_scale=q_scale)

    def gnstats_ok(self):
        """True when forward_gnstats serves this layer: a bf16 3x3 / stride-1 convolution into GroupNorm(32, 256)'s 256 channels."""
        return (not self.fp8) and self.k == 3 and self.stride == 1 and self.pad == 1 and self.cout == 256 and self.cin % 64 == 0

    def forward_gnstats(self, x, gin, gout, y, part):
        """forward() of a tower convolution that also leaves GroupNorm's per-patch statistics in `part` (ops.conv2d_fwd_gnstats)."""
        return ops.conv2d_fwd_gnstats(self.desc(gin, gout), x, self.w_fwd, self.b, y, part)

    def dgrad(self, g, gin, gout, dx, first=True, mask=None, add_after=None, maskbits=None, g8=None, dx8=None, q_scale=1.0,
              add_before=None, sparse=False):
        """dx (+)= conv^T(g); returns True when the launch also wrote the e5m2 twin dx8 (= dx * q_scale, q_scale = the CONSUMER's
        gradient scale).  first=False accumulates onto dx (pre-mask); mask = forward activation whose
        ReLU gates dx (maskbits: the same gate bit-packed, written by the producing forward launch); add_after = tensor added
        after masking (P6: gradient that bypasses the ReLU)."""
        flags, add = 0, None
        if add_after is not None:
            flags |= ops.EPI_ADD_AFTER
            add = add_after
        elif add_before is not None:                 # dx = (conv^T(g) + add_before) * mask: the identity skip's gradient joins here
            flags |= ops.EPI_ADD_BEFORE
            add = add_before
        elif not first:
            flags |= ops.EPI_ADD_BEFORE
            add = dx
            if sparse and self.stride > 1:       # strided shortcut accumulating in place: pixels no tap reaches keep their (final) value
                flags |= ops.EPI_SPARSE
        if mask is not None or maskbits is not None:
            flags |= ops.EPI_MASK
        if maskbits is not None:
            mask = None
        d = self.desc(gin, gout)
        if self.amax_slot is not None and self.probe_ctl[0] and (self.fp8_1x1_dgrad or self.fp8_dgrad):
            ops.absmax_bf16(g, self.amax_slot)
        if self.fp8_1x1_dgrad and g8 is not None and ops.conv1x1_fp8_ok(d, 1):
            ops.conv1x1_fp8(d, 1, g8, self.w_q8t, self.w_scale8t, None, dx, add=add, mask=mask, maskbits=maskbits, y8=dx8,
                            q_scale=q_scale, flags=flags)
            return dx8 is not None
        if self.fp8_dgrad and maskbits is None:
            # g8: the e5m2 twin of g (g * grad_scale) when the producing launch wrote one, else a cast pass; dx8: twin of dx to write
            gq = g8 if g8 is not None else ops.quantize_bf8(g, self.grad_scale, self.q8_scratch(g.numel())[: g.numel()])
            ops.conv2d_dgrad_fp8(d, gq, self.w_q8t, self.w_scale8t, dx, add=add, mask=mask, flags=flags, dx8=dx8, q_scale=q_scale)
            return dx8 is not None
        if dx8 is not None and not ops.dense_1x1_bits_ok(d):
            dx8 = None               # only the dense 1x1 kernel writes a twin on the bf16 path: the caller must not mark it ready
        ops.conv2d_dgrad(d, g, self.w_dgrad, dx, add=add, mask=mask, flags=flags, maskbits=maskbits, dx8=dx8, q_scale=q_scale)
        return dx8 is not None

    def dgrad_writes_twin(self, gin, gout):
        """True when dgrad(..., dx8=t) fills t: the fp8 patch kernel and the dense 1x1 kernel do, the other bf16 kernels do not."""
        return self.fp8_dgrad or ops.dense_1x1_bits_ok(self.desc(gin, gout))

    def wgrad(self, x, g, gin, gout, ws, colsum_ws=None, x8=None, g8=None, queue=None):
        """x8 / g8: the e4m3 twin of x (x * act_scale) and the e5m2 twin of g (g * grad_scale) when their producers wrote them: the
        weight gradient then runs on the one-byte kernel (the bias gradient stays a column sum of the bf16 g)."""
        d = self.desc(gin, gout)
        if self.fp8_wgrad and x8 is not None and g8 is not None:
            ops.conv2d_wgrad_fp8(d, x8, g8, 1.0 / (self.act_scale * self.grad_scale), self.gw, ws, row_scale=self.row_scale)
            if self.gb is not None:
                ops.colsum_bf16(g, g.shape[0], self.cout, self.gb, colsum_ws)
            return
        if queue is not None:        # ops.WgradQueue: the reduce waits for the caller's flush (ws stays untouched until then)
            queue.wgrad(d, x, g, self.gw, self.gb, ws, row_scale=self.row_scale)
        elif self.gb is not None:    # weight + bias gradient in one entry point (fused in the 3x3 patch kernel; colsum_ws is unused)
            ops.conv2d_wgrad_bias(d, x, g, self.gw, self.gb, ws, row_scale=self.row_scale)
        else:
            ops.conv2d_wgrad(d, x, g, self.gw, ws, row_scale=self.row_scale)

    def thin_backward_ok(self, gin: Geom):
        """True when bd_conv1x1_thin_bwd covers this layer on geometry gin: 1x1 / stride 1, 256 -> 16 (padded) channels, bf16, no BN scale,
        every pixel row of the buffer (all levels of the pyramid, contiguous)."""
        return (self.k == 1 and self.stride == 1 and self.pad == 0 and self.cin == 256 and self.cout == 16 and self.trainable
                and self.row_scale is None and self.gb is not None and not (self.fp8 or self.fp8_1x1)
                and sum(h * w for h, w in zip(gin.H, gin.W)) == gin.pix_per_img)

    def thin_forward(self, x, gin: Geom, y):
        """y = conv(x) + bias on the thin-layer kernel (fp32 master weights rounded to bf16 in the kernel, as pack() does)."""
        ops.conv1x1_thin_fwd(x, self.w, self.b, gin.pixels, self.cin, self.cout, y)

    def thin_backward(self, x, g, gin: Geom, dx, ws):
        """dx = (x > 0) * conv^T(g), dW, dbias in one pass over x (x = the ReLU output this layer reads).  Same results as
        wgrad(x, g) + dgrad(g, mask=x) up to fp32 summation order."""
        ops.conv1x1_thin_bwd(x, g, self.w, gin.pixels, self.cin, self.cout, dx, self.gw, self.gb, self.cout_real, ws)

    def wgrad_ws_bytes(self, gin, gout):
        d = self.desc(gin, gout)
        n = ops.conv2d_wgrad_bias_workspace_bytes(d) if self.gb is not None else ops.conv2d_wgrad_workspace_bytes(d)
        return max(n, ops.conv2d_wgrad_fp8_workspace_bytes(d)) if self.fp8_wgrad else n


class FusedPredConv(ConvLayer):
    """Several reference convs that read the same input, fused along Cout into one launch (FCOS bbox_pred [4] + ctrness [1]
    -> 5 rows, padded to 8).  Parameters are bound from / exported to the reference's separate tensors."""

    def __init__(self, name, parts, cin, k, stride, pad, device, cout_pad):
        self.parts = parts   # [(reference conv name, cout)]
        super().__init__(name, cin, sum(c for _, c in parts), k, stride, pad, device, has_bias=True, trainable=True, cout_pad=cout_pad)

    def bind(self, arena, params):
        merged = dict(params)
        merged[self.name + ".weight"] = np.concatenate([params[n + ".weight"] for n, _ in self.parts], 0)
        merged[self.name + ".bias"] = np.concatenate([params[n + ".bias"] for n, _ in self.parts], 0)
        super().bind(arena, merged)

    def export(self, out):
        o = 0
        for n, c in self.parts:
            out[n + ".weight"] = self.w[o:o + c].permute(0, 3, 1, 2).contiguous().cpu().numpy()
            out[n + ".bias"] = self.b[o:o + c].cpu().numpy().copy()
            o += c

    def export_grad(self, out):
        o = 0
        for n, c in self.parts:
            out[n + ".weight"] = self.gw[o:o + c].permute(0, 3, 1, 2).contiguous().cpu()
            out[n + ".bias"] = self.gb[o:o + c].cpu().clone()
            o += c


class FCLayer(ConvLayer):
    """megengine.module.Linear (or several that read the same input, fused along the output dimension) as a 1x1 convolution
    whose "pixels" are the rows.  The reference flattens RoI features as (c, h, w) (layers/head/rcnn.py:59); the HIP
    RoIAlign writes (h, w, c), so for `in_chw` layers the master weight is kept with permuted columns -- an elementwise SGD
    does not care -- and permuted back on export."""

    def __init__(self, name, cin, cout, device, in_chw=None, parts=None, cout_pad=None):
        self.in_chw = in_chw
        self.parts = parts          # [(reference linear name, out features)]
        super().__init__(name, cin, cout, 1, 1, 0, device, has_bias=True, trainable=True, cout_pad=cout_pad)

    def _names(self):
        return [n for n, _ in self.parts] if self.parts else [self.name]

    def _to_ours(self, w):
        if self.in_chw:
            c, h, wd = self.in_chw
            w = w.reshape(-1, c, h, wd).permute(0, 2, 3, 1).reshape(-1, c * h * wd)
        return w.contiguous()

    def _to_ref(self, w):
        if self.in_chw:
            c, h, wd = self.in_chw
            w = w.reshape(-1, h, wd, c).permute(0, 3, 1, 2).reshape(-1, c * h * wd)
        return w.contiguous()

    def bind(self, arena, params):
        w = np.concatenate([np.asarray(params[n + ".weight"], np.float32) for n in self._names()], 0)
        b = np.concatenate([np.asarray(params[n + ".bias"], np.float32) for n in self._names()], 0)
        w = self._to_ours(torch.from_numpy(w)).numpy().reshape(self.cout_real, self.cin, 1, 1)
        super().bind(arena, {self.name + ".weight": w, self.name + ".bias": b})

    def _split(self, w, b, out, to_numpy):
        w = self._to_ref(w[: self.cout_real].reshape(self.cout_real, self.cin))
        b = b[: self.cout_real]
        o = 0
        for n, c in (self.parts or [(self.name, self.cout_real)]):
            wn, bn = w[o:o + c].cpu(), b[o:o + c].cpu()
            out[n + ".weight"] = wn.numpy().copy() if to_numpy else wn.clone()
            out[n + ".bias"] = bn.numpy().copy() if to_numpy else bn.clone()
            o += c

    def export(self, out):
        self._split(self.w, self.b, out, True)

    def export_grad(self, out):
        self._split(self.gw, self.gb, out, False)
