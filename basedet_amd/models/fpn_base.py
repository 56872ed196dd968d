"""Shared ResNet + FPN trunk of the one-stage detectors (RetinaNet, FCOS) on the HIP path.

Same protocol as the reference's BaseNet (models/base_net.py:50-71): ``model(batch)`` in training mode returns
``{"total_loss", "cls_loss", "reg_loss"}``; the batch dict is the collator contract
(data/collators/pad_collator.py:38-49): ``data`` (N,3,H,W), ``gt_boxes`` (N,G,5), ``im_info`` (N,5).
There is no autograd: ``get_losses`` runs the forward kernels and the fused loss fwd+bwd kernels, ``backward()``
runs the explicit dgrad/wgrad schedule in reverse order and leaves fp32 gradients in the parameter arena.

Gradient convention: the gradient buffer of a post-ReLU activation T holds dL/d(pre-activation) (already masked
by T > 0).  When T has several consumers, all but the last dgrad accumulate unmasked and the last one applies the
mask in its epilogue -- so ReLU backward, residual fan-in and FrozenBN never cost a separate pass over HBM.
"""
import math
from contextlib import nullcontext as _nullcontext

import numpy as np
import torch

from .. import ops
from ..ops import Geom
from . import params as P
from .engine import ConvLayer, FCLayer, ParamArena


def _round_up(v, m):
    return (v + m - 1) // m * m


class _Plan:
    """Shape-dependent buffers (activations, gradients, anchors, workspaces) for one (N, Hp, Wp)."""


class VecParam:
    """A 1-D trainable fp32 parameter that is not a conv weight (GroupNorm affine, FCOS level scales)."""

    def __init__(self, name, numel):
        self.name, self.numel = name, numel
        self.w = self.g = None

    def reserve(self, arena):
        self._i = arena.reserve(self.name, (self.numel,))

    def bind(self, arena, params):
        self.w = arena.view("w", self._i)
        self.g = arena.view("g", self._i)
        self.w.copy_(torch.from_numpy(np.asarray(params[self.name], np.float32).reshape(-1)))

    def export(self, out):
        out[self.name] = self.w.cpu().numpy().copy()


class FPNDetector:
    """Backbone + FPN forward/backward and the BaseNet module protocol; heads and losses live in subclasses."""

    TOP_BLOCK = "p6p7"      # LastLevelP6P7 (RetinaNet / FCOS); "pool" = FPNP6 (Faster R-CNN)

    def __init__(self, cfg, params=None, device="cuda", seed=0):
        self.cfg = cfg
        self.device = torch.device(device)
        m = cfg.MODEL
        self.training = True
        self.num_classes = cfg.DATA.NUM_CLASSES
        self.strides = list(m.FPN.STRIDES)
        self.fpn_ch = m.FPN.OUT_CHANNELS
        self.freeze_at = m.BACKBONE.FREEZE_AT
        self.img_mean, self.img_std = list(m.BACKBONE.IMG_MEAN), list(m.BACKBONE.IMG_STD)
        if params is None:
            params = self.init_params(cfg, seed)
        self._build_layers(params)
        self._plans = {}
        self._cur = None
        self.extra_meter = {}
        # weight-gradient kernels run on a side stream, concurrently with the dgrad chain they do not feed: tails and
        # barrier bubbles of one kernel are filled by the other (set False to serialise, e.g. for per-kernel timing)
        self.async_wgrad = True
        # WGRAD_QUEUE: "layer" (default again since round 5) = one fixed-order reduce per layer right behind its partial-sum kernel: the slabs
        # are still in the Infinity Cache when they are read back; "bucket" (round 4's default) = the reduces of a gradient bucket (head /
        # fpn / layer4 / layer3 / layer2) in ONE launch (bd_wgrad_queue_*: 5 reduce launches per step instead of 60); an integer =
        # additionally flush whenever that many bytes of partial sums are pending.  Same bits in every mode (tests/test_wgrad_queue_gpu.py).
        # Measured, alternating on one box (profiles/r05_workloads.txt, r05_queue_ab.txt; round 4 had read the same sign and called it
        # neutral): layer 635.8 / 637.0 / 636.3 img/s, bucket 632.0 / 630.9 / 635.1 -- and 647.4 / 643.7 against 643.2 / 639.5 on two other boxes.
        self.wgrad_queue_mode = m.get("WGRAD_QUEUE", "layer")
        self._wq = None
        self._wq_need = {}                  # (layer name, full geometry) -> workspace bytes
        # ONE partial-sum arena per model, as large as the largest flush interval (gradient bucket) seen so far: a flush's reduce and
        # every later partial-sum kernel run on the same stream, so the slices are re-used from offset 0 after each flush
        self._wq_arena = None
        self._wq_off = self._wq_pending = self._wq_peak = 0
        self.use_mask_bits = True          # bit-packed ReLU gates for the wide 1x1 data gradients (False: bf16 activations as masks)
        self._wstream = torch.cuda.Stream() if (torch.cuda.is_available() and self.device.type == "cuda") else None
        self._tstream = torch.cuda.Stream() if self._wstream is not None else None      # P6/P7 top-block dgrads

    # ------------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------------
    def _build_layers(self, params):
        dev = self.device
        m = self.cfg.MODEL
        self.arena = ParamArena(dev)
        bu = "backbone.bottom_up"
        self.blocks = P.resnet_conv_table(m.BACKBONE.NAME)
        self.convs = {}

        def add(name, cin, cout, k, stride, pad, bn=None, bias=False, trainable=True, cout_pad=None):
            c = ConvLayer(name, cin, cout, k, stride, pad, dev, bn_prefix=bn, has_bias=bias, trainable=trainable, cout_pad=cout_pad)
            self.convs[name] = c
            return c

        # stem (frozen when FREEZE_AT >= 1; the dedicated 7x7 kernel is forward-only)
        assert self.freeze_at >= 1, "the 7x7 stem kernel is forward-only: FREEZE_AT must be >= 1 (reference default 2)"
        self.stem_packed = torch.empty((64, 7, 8, 4), dtype=torch.bfloat16, device=dev)

        for blk in self.blocks:
            pre = blk["prefix"]
            tr = not (blk["layer"] == 1 and self.freeze_at >= 2)
            blk["trainable"] = tr
            if blk["kind"] == "bottleneck":
                blk["convs"] = [
                    add(pre + ".conv1", blk["cin"], blk["ch"], 1, 1, 0, bn=pre + ".bn1", trainable=tr),
                    add(pre + ".conv2", blk["ch"], blk["ch"], 3, blk["stride"], 1, bn=pre + ".bn2", trainable=tr),
                    add(pre + ".conv3", blk["ch"], blk["cout"], 1, 1, 0, bn=pre + ".bn3", trainable=tr),
                ]
            else:
                blk["convs"] = [
                    add(pre + ".conv1", blk["cin"], blk["ch"], 3, blk["stride"], 1, bn=pre + ".bn1", trainable=tr),
                    add(pre + ".conv2", blk["ch"], blk["cout"], 3, 1, 1, bn=pre + ".bn2", trainable=tr),
                ]
            blk["ds"] = add(pre + ".downsample.0", blk["cin"], blk["cout"], 1, blk["stride"], 0, bn=pre + ".downsample.1",
                            trainable=tr) if blk["has_ds"] else None

        self.fpn_stages = [int(f[-1]) for f in m.BACKBONE.OUT_FEATURES]          # [3, 4, 5]
        ch = self.fpn_ch
        self.lateral, self.output = {}, {}
        for s, ci in zip(self.fpn_stages, m.BACKBONE.OUT_FEATURE_CHANNELS):
            self.lateral[s] = add(f"backbone.fpn_lateral{s}", ci, ch, 1, 1, 0, bias=True)
            self.output[s] = add(f"backbone.fpn_output{s}", ch, ch, 3, 1, 1, bias=True)
        if self.TOP_BLOCK == "p6p7":
            self.p6 = add("backbone.top_block.p6", m.FPN.TOP_BLOCK_IN_CHANNELS, ch, 3, 2, 1, bias=True)
            self.p7 = add("backbone.top_block.p7", ch, ch, 3, 2, 1, bias=True)
        self.vparams = {}
        self._build_head(add, params)

        for c in list(self.convs.values()) + list(self.vparams.values()):
            c.reserve(self.arena)
        self.arena.allocate()
        # BASELINE config 5: fp8-e4m3 weights (one scale per output channel) for the forward of the 3x3 convolutions -- where a
        # quantised copy of the input is read nine times; the HBM-bound 1x1 layers and the whole backward pass stay bf16
        self.fuse_stem_pool = bool(m.get("FUSE_STEM_POOL", True))
        # frozen bottleneck blocks (layer1 under FREEZE_AT = 2) in one launch each: the two mid tensors and the residual re-read never
        # reach HBM (bd_bottleneck_fwd; False: the three / four bd_conv2d_fwd launches, kept as the parity reference)
        self.fuse_frozen_blocks = bool(m.get("FUSE_FROZEN_BLOCKS", True))
        self.sparse_shortcut_grad = bool(m.get("SPARSE_SHORTCUT_GRAD", True))   # False: the shortcut's data gradient as a full-resolution pass (A/B)
        self.weight_dtype = m.get("WEIGHT_DTYPE", "bf16")
        # FP8_DGRAD (default True again since round 3): e5m2 gradients x e4m3 weights for the data gradients of the fp8 layers, +2.5-3.5 % on
        # the step.  Round 2 had to make it opt-in: with ONE delayed scale for all layers R101 at the batch-32 learning rate left the
        # finite range between steps 620 and 1 020 of the repeated-batch run.  Round 3: one scale per backward phase (FP8_SCALE_GROUPS)
        # with two binades of headroom below e5m2's maximum and a four-probe history -- see _fp8_probe_end and DESIGN.md (a21).
        fp8_dgrad_default = True
        # Initial (pre-probe) scale of the e5m2 gradients.  Every loss is normalised by a count that grows with the batch (num_fg, sample
        # counts), so the gradients shrink like 1 / batch: a fixed 4 096 put the head's gradients of a 32-image batch next to e5m2's
        # subnormals for the first FP8_AMAX_DELAY steps (tests/test_bench_batch_gpu.py: cls_subnet weight gradient 51 % off the batch-2
        # one).  4 096 was tuned on two images; the default follows the batch in powers of two until the first probe takes over.
        fp8_scale0 = 4096.0 * 2.0 ** max(0, int(round(np.log2(max(1, int(m.get("BATCHSIZE", 2))) / 2.0))))
        self._q8 = {}
        # e5m2 twins of gradients written by the producing launch (False: every fp8 data gradient casts its input in a pass; a test knob)
        self.fp8_grad_twins = bool(m.get("FP8_GRAD_TWINS", True))
        if self.weight_dtype == "fp8_e4m3":
            side = {id(getattr(self, n)) for n in ("p6", "p7") if hasattr(self, n)}
            for c in self.convs.values():
                # the staggered fp8 patch kernel serves 3x3 / stride 1 with Cout > 128 (1.6x the bf16 kernel); the top block's stride-2
                # convolutions go through the generic fp8 kernel (faster than their bf16 launches, and they complete the pyramid's e4m3
                # twin); narrower or strided backbone layers are FASTER on their bf16 kernels and stay there
                if (c.k == 3 and c.cin % 16 == 0 and not isinstance(c, FCLayer)
                        and ((c.stride == 1 and c.cout > 128) or id(c) in side)):
                    key = "side" if id(c) in side else "main"          # P6 / P7 run on a side stream: their own scratch
                    c.enable_fp8(lambda n, key=key: self._q8_buf(key, n), m.get("FP8_ACT_SCALE", 1.0),
                                 dgrad=bool(m.get("FP8_DGRAD", fp8_dgrad_default)), grad_scale=m.get("FP8_GRAD_SCALE", fp8_scale0),
                                 # FP8_WGRAD: 0 = bf16 weight gradients, 1 = the one-byte kernel (bd_conv2d_wgrad_fp8) for the bias-free
                                 # layers (backbone conv2), 2 (default since round 4) = also the towers.  Exact on representable inputs;
                                 # 1.4x the bf16 ring kernel per head-tower launch.  Round 2 kept it opt-in (no gain in the step then,
                                 # and a repeated-batch run under the static scale lost convergence); under per-group delayed scales and
                                 # stochastic rounding: R101 batch 32 same box 487.6 / 487.6 img/s at 0, 494.5 / 496.4 at 2; ten of ten
                                 # seeds through 1 500 repeated-batch steps (profiles/r04_fp8_stability_wgrad.txt; bf16 and the
                                 # FP8_WGRAD = 0 form: nine of ten each); whole-model gradient cosine vs bf16 0.9814 (0.9818 at 0) at
                                 # 2 x 800 x 1344 (tests/test_r101_gpu.py)
                                 wgrad=(int(m.get("FP8_WGRAD", 2)) >= (2 if c.has_bias else 1)) and bool(m.get("FP8_DGRAD", fp8_dgrad_default)))
            # the bottleneck 1x1s around an fp8 3x3 (res4 / res5 blocks after the first) on one-byte operands.  In isolation the reducing
            # direction (conv1 forward, conv3's data gradient: the input is most of the bytes) is 1.5 - 1.6x faster than its bf16 launch
            # and the expanding one about even; in the step the extra twins the neighbouring launches must write take most of it back:
            # R101 batch 32, one box: 461.3 img/s without, 463.9 with both directions (the default), 458.5 with the reducing one only
            if bool(m.get("FP8_1X1", True)):
                for blk in self.blocks:
                    if blk["kind"] == "bottleneck" and blk["convs"][1].fp8 and blk["convs"][1].stride == 1:
                        for c in (blk["convs"][0], blk["convs"][2]):
                            if c.cin % 32 == 0 and c.cout % 32 == 0:
                                c.enable_fp8_1x1(m.get("FP8_ACT_SCALE", 1.0), dgrad=bool(m.get("FP8_DGRAD", fp8_dgrad_default)),
                                                 grad_scale=m.get("FP8_GRAD_SCALE", fp8_scale0),
                                                 expanding=bool(m.get("FP8_1X1_EXPANDING", True)))
        else:
            assert self.weight_dtype == "bf16", self.weight_dtype
        # Delayed scaling of the e5m2 gradients (the reference's hook for this is the AMP GradScaler, solver/default_solver.py:66-76): one
        # scale for all fp8 data gradients (twins pass from layer to layer, so the layers must agree on it), re-derived every
        # FP8_AMAX_INTERVAL steps from max |g| over the gradients those launches consume -- measured by bd_absmax_bf16 on the probe step,
        # copied to the host asynchronously and applied FP8_AMAX_DELAY steps later, after that step's data gradients and before its
        # weight repack, so that quantisation and the folded 1 / scale of the packed weights always agree.  A static scale underflows
        # once training has shrunk the gradients: 4 096 diverged after ~1 500 steps of the repeated-batch run, 65 536 did not.
        self._fp8_grad_layers = [c for c in self.convs.values() if c.fp8_dgrad or c.fp8_1x1_dgrad or c.fp8_wgrad]
        if self._fp8_grad_layers:
            for c in self.convs.values():             # one scale everywhere at the start (a twin's producer reads it off the consumer's layer object)
                c.grad_scale = float(m.get("FP8_GRAD_SCALE", fp8_scale0))
        self.fp8_delayed_scaling = bool(m.get("FP8_DELAYED_SCALING", True)) and bool(self._fp8_grad_layers) and self.device.type == "cuda"
        # Stochastic rounding of those gradients (bd_conv_desc.sr_seed): round-to-nearest e5m2 repeats the same error on the
        # same value every step, which a repeated batch turns into a drift (the long repeated-batch runs)
        self.fp8_stochastic_rounding = bool(m.get("FP8_STOCHASTIC_ROUNDING", True)) and bool(self._fp8_grad_layers) and self.device.type == "cuda"
        self.fp8_amax_interval = int(m.get("FP8_AMAX_INTERVAL", 10))
        self.fp8_amax_delay = int(m.get("FP8_AMAX_DELAY", 4))
        self.fp8_amax_history = max(1, int(m.get("FP8_AMAX_HISTORY", 4)))         # probes whose maximum sets the scale
        # max |g| * scale lands in (2^(t-1), 2^t]; e5m2 tops out at 1.75 * 2^15 and everything above is CLAMPED.  One global scale: t = 15 (R50:
        # 12 and the static 4 096 diverged in the 2 020-step run, 13 - 15 did not; R101 at batch 32: 14 diverged before step 1 020): only
        # the head sits at the top of the range, every other layer has binades of headroom.  Per-group scales put EVERY group at the top,
        # and a group whose gradients grow between two probes then saturates: R101 batch 32 at t = 15 had layer3 at 92 672 = 1.6 x the
        # maximum at step 100 and left the finite range before step 400, while t = 12 (0.3035 after 1 500 steps) and an eight-probe
        # history at t = 15 (0.3306) both ran through (profiles/r03_fp8_scale_groups.txt).  Round 3 shipped t = 13 with a four-probe
        # history, a combination that had NOT run through (seed 0 diverged on it, same file, section 2): the default is t = 12 with the
        # four-probe history, the combination that did; the round-4 seed matrix is profiles/r04_fp8_stability.txt.
        self.fp8_amax_target = float(m.get("FP8_AMAX_TARGET_LOG2", 15.0 if str(m.get("FP8_SCALE_GROUPS", "group")) == "global" else 12.0))
        # Granularity of the delayed scale (round 3).  One scale for all layers had to span the 2^9.7 spread between max |g| at the head and
        # at the backbone's conv1 layers (scripts/exp/fp8_amax_spread.py): with the head's maximum at 2^15 the backbone's gradients sat
        # ten binades lower and their small values flushed to zero -- R101 at the batch-32 learning rate left the finite range.  "group"
        # (default) keeps one scale per backward phase (head / fpn / layer4 / layer3 / layer2: the all-reduce buckets), "layer" one
        # per layer, "global" the round-2 behaviour.  Every twin is written with its CONSUMER's scale (q_scale at each hand-off) and a
        # layer's transposed fp8 weights fold in 1 / its own scale, so any partition is consistent.
        self.fp8_scale_groups = str(m.get("FP8_SCALE_GROUPS", "group"))
        assert self.fp8_scale_groups in ("global", "group", "layer"), self.fp8_scale_groups
        self._fp8_t, self._amax_pending, self._amax_prev = 0, None, {}
        self._fp8_staged = None                  # {conv: scale} waiting for the next weight repack (optimizer step)
        self.fp8_scale_log = []
        self.fp8_group_scales = {}
        if self.fp8_delayed_scaling:
            n = len(self._fp8_grad_layers)
            self._amax_dev = torch.zeros(n, dtype=torch.float32, device=self.device)
            self._amax_host = torch.zeros(n, dtype=torch.float32).pin_memory()
            self._probe_ctl = [False]
            for i, c in enumerate(self._fp8_grad_layers):
                c.amax_slot, c.probe_ctl = self._amax_dev[i:i + 1], self._probe_ctl
        self._bind_params(params)

    def _fp8_probe_begin(self):
        if self.fp8_stochastic_rounding:          # this step's e5m2 quantisers: a new hash seed per step (reset at the end of backward)
            ops.fp8_set_stochastic_rounding(((self._fp8_t + 1) * 2654435761 + 0x9E3779B9) | 1)
        if not self.fp8_delayed_scaling:
            self._fp8_t += 1
            return False
        t = self._fp8_t
        self._fp8_t += 1
        if self._amax_pending is None and t % self.fp8_amax_interval == 0:
            self._amax_dev.zero_()
            self._probe_ctl[0] = True
            return True
        return False

    def _fp8_scale_key(self, c):
        if self.fp8_scale_groups == "global":
            return "all"
        if self.fp8_scale_groups == "layer":
            return "fpn_output" if "fpn_output" in c.name else c.name      # the output convolutions all read ONE twin of dL/dP
        n = c.name
        if n.startswith(("head.", "rpn.", "rcnn.")):
            return "head"
        if "fpn_" in n or "top_block" in n:
            return "fpn"
        return n.split(".")[2] if n.startswith("backbone.bottom_up.") else "head"

    def _fp8_apply_staged(self):
        """New gradient scales take effect HERE, together with the weight repack that folds 1 / scale into the transposed fp8 weights
        (repack_trainable, i.e. the optimizer step): a second backward() without a step keeps quantisers and weights consistent."""
        st, self._fp8_staged = self._fp8_staged, None
        if st:
            for c, sc in st.items():
                c.grad_scale = sc

    def _fp8_probe_end(self, probing):
        if self.fp8_stochastic_rounding:
            ops.fp8_set_stochastic_rounding(0)
        if not self.fp8_delayed_scaling:
            return
        t = self._fp8_t - 1
        if probing:
            self._probe_ctl[0] = False
            from .. import comm as _comm
            cm = _comm.get_comm()
            if cm is not None and cm.world > 1:       # every rank quantises with the same scales: max |g| over the ranks
                cm.allreduce_async(self._amax_dev, [torch.cuda.current_stream()], "max")
                cm.wait()
            self._amax_host.copy_(self._amax_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._amax_pending = (ev, t)
        elif self._amax_pending is not None and t >= self._amax_pending[1] + self.fp8_amax_delay:
            ev, t0 = self._amax_pending
            ev.synchronize()                      # long done: the host runs a few steps ahead of the device, not FP8_AMAX_DELAY + the queue
            self._amax_pending = None
            am = self._amax_host.numpy().astype(np.float64)
            keys = {}
            for i, c in enumerate(self._fp8_grad_layers):
                k = self._fp8_scale_key(c)
                if np.isfinite(am[i]):
                    keys[k] = max(keys.get(k, 0.0), float(am[i]))
            scales = {}
            # diagnostic: where the probe's largest value sat in e5m2's range under the scale it was quantised with (> 57 344: clamped)
            cur = {self._fp8_scale_key(c): c.grad_scale for c in self._fp8_grad_layers}
            self.fp8_last_fill = {k: a * cur[k] for k, a in keys.items()}
            for k, amax in keys.items():
                if amax > 0.0:
                    hist = self._amax_prev.setdefault(k, [])          # probe history: a scale never chases a single small reading
                    hist.append(amax)
                    del hist[:-self.fp8_amax_history]
                    eff = max(hist)
                    scales[k] = float(2.0 ** min(max(np.floor(self.fp8_amax_target - np.log2(eff)), -16.0), 40.0))
            if scales:
                staged = {}
                for c in self.convs.values():         # every layer of a group (a twin's producer reads the scale off its CONSUMER's layer object)
                    k = self._fp8_scale_key(c)
                    if k in scales:
                        staged[c] = scales[k]
                self._fp8_staged = staged
                self.fp8_group_scales.update(scales)
                top = max(keys.values())
                self.fp8_scale_log.append((t0, t, top, min(scales.values())))

    def _bind_params(self, params):
        """(Re)load every parameter from a reference-layout dict (name -> numpy) and refresh the packed bf16 copies."""
        dev = self.device
        bu = "backbone.bottom_up"
        self.stem_w = torch.from_numpy(np.asarray(params[bu + ".conv1.weight"], np.float32)).permute(0, 2, 3, 1).contiguous().to(dev)
        g, b = params[bu + ".bn1.weight"], params[bu + ".bn1.bias"]
        mu, var = params[bu + ".bn1.running_mean"], params[bu + ".bn1.running_var"]
        sc = g / np.sqrt(var + 1e-5)
        self.stem_scale = torch.from_numpy(sc.astype(np.float32)).to(dev)
        self.stem_shift = torch.from_numpy((b - mu * sc).astype(np.float32)).to(dev)
        self._pack_table = None                 # row_scale tensors are re-created by bind()
        for c in list(self.convs.values()) + list(self.vparams.values()):
            c.bind(self.arena, params)
        self._bn_params = {k: np.asarray(v, np.float32).copy() for k, v in params.items() if (".bn" in k or "downsample.1" in k)}
        self.repack_weights()

    def _q8_buf(self, key, nbytes):
        """Scratch for the e4m3 copy of a convolution's input, one per stream (grown on demand; stream-ordered reuse)."""
        t = self._q8.get(key)
        if t is None or t.numel() < nbytes:
            t = self._q8[key] = torch.empty((int(nbytes),), dtype=torch.uint8, device=self.device)
        return t

    def load_weights(self, weights, strict=False):
        """BaseNet.load_weights (models/base_net.py:83-89) -> utils/checkpoint.py load_matched_weights: `weights` is a dict
        or the path of a .pkl / .npz checkpoint; names are matched exactly, then by suffix, then by shape."""
        from ..utils.checkpoint import load_matched_weights
        return load_matched_weights(self, weights, strict)

    def repack_weights(self):
        """Refresh the bf16 packed copies from the fp32 masters (after load_weights / every optimizer step)."""
        ops.stem_weight_pack(self.stem_w, self.stem_scale, self.stem_packed)
        for c in self.convs.values():
            c.pack()

    def repack_trainable(self):
        """All trainable convs in ONE launch (bd_weight_pack_multi); the table holds raw pointers into the arena and the packed
        tensors, which never move after _build_layers."""
        self._fp8_apply_staged()
        if getattr(self, "_pack_table", None) is None:
            ent = [(c.w, c.row_scale, c.w_fwd, c.w_dgrad, c.cout, c.k * c.k, c.cin) for c in self.convs.values() if c.trainable]
            self._pack_table = ops.build_pack_table(ent, self.device)
        ops.weight_pack_multi(self._pack_table)
        for c in self.convs.values():
            if (c.fp8 or c.fp8_1x1 or c.fp8_1x1_dgrad) and c.trainable:
                c.pack_fp8()

    # reference module protocol ------------------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def __call__(self, inputs):
        return self.forward(inputs)

    def forward(self, inputs):
        """BaseNet.forward (models/base_net.py:50-54)."""
        if self.training:
            return self.get_losses(inputs)
        return self.inference(inputs)

    def state_dict(self):
        out = {}
        for c in list(self.convs.values()) + list(self.vparams.values()):
            c.export(out)
        out["backbone.bottom_up.conv1.weight"] = self.stem_w.permute(0, 3, 1, 2).contiguous().cpu().numpy()
        out.update({k: v.copy() for k, v in self._bn_params.items()})
        return out

    def debug_activations(self):
        """Stored activations of the last forward as NCHW fp32 CPU tensors, keyed like oracle/model.py `_act`
        (parity tests inject them into the oracle so that both backward passes see identical ReLU gates)."""
        pl = self._cur
        N = pl.N
        out = {}

        def nchw(t, g, c=None):
            v = t.float().cpu().view(N, g.H[0], g.W[0], -1).permute(0, 3, 1, 2).contiguous()
            return v if c is None else v[:, :c].contiguous()

        def lvl(t, i, c=None):
            g = pl.pyr
            v = t.float().cpu().view(N, g.pix_per_img, -1)[:, g.off[i]: g.off[i] + g.H[i] * g.W[i]]
            v = v.reshape(N, g.H[i], g.W[i], -1).permute(0, 3, 1, 2).contiguous()
            return v if c is None else v[:, :c].contiguous()

        out["pool"] = nchw(pl.pool_out, pl.g_pool)
        for blk, b in zip(self.blocks, pl.blk):
            pre = blk["prefix"]
            for i, (t, g) in enumerate(zip(b.mids, b.mid_geo)):
                out[f"{pre}.a{i}"] = nchw(t, g)
            if b.idt is not None:
                out[pre + ".idt"] = nchw(b.idt, b.gout)
            out[pre + ".out"] = nchw(b.out, b.gout)
        for s in self.fpn_stages:
            out[f"lat{s}"] = nchw(pl.lat[s], pl.blk[pl.res[s]].gout)
        for i in range(pl.pyr.nlev):
            out[f"P{self.fpn_stages[0] + i}"] = lvl(pl.P, i)
        self._debug_head(pl, out, lvl)
        return out

    def reference_grads(self):
        """Gradients of the last backward() keyed by the reference's parameter names, in the reference's layouts
        (conv weights OIHW, padding rows dropped) -- what megengine's GradManager would hand to the optimizer."""
        out = {}
        for c in self.convs.values():
            if not c.trainable:
                continue
            c.export_grad(out)
        for v in self.vparams.values():
            out[v.name] = v.g.detach().cpu().clone()
        return out

    def state_dict_trainable_names(self):
        """Trainable parameters under the reference's names (what DetSolver.params would collect)."""
        return list(self.reference_grads_names())

    def reference_grads_names(self):
        names = []
        for c in self.convs.values():
            if not c.trainable:
                continue
            parts = getattr(c, "parts", None)
            for n in ([p[0] for p in parts] if parts else [c.name]):
                names.append(n + ".weight")
                if c.has_bias and not c.bn_prefix:
                    names.append(n + ".bias")
        names += [v.name for v in self.vparams.values()]
        return names

    def trainable_parameter_names(self):
        return [e[0] for e in self.arena.entries]

    # ------------------------------------------------------------------------------------------------
    # shape plan
    # ------------------------------------------------------------------------------------------------
    def _plan(self, N, Hp, Wp):
        key = (N, Hp, Wp)
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        dev = self.device
        pl = _Plan()
        pl.N, pl.Hp, pl.Wp = N, Hp, Wp
        bf = dict(dtype=torch.bfloat16, device=dev)

        def act(g, c):
            return torch.empty((g.pixels, c), **bf)

        pl.x_halo = torch.empty((N, Hp + 6, Wp + 8, 4), **bf)
        g2 = ops.single(N, Hp // 2, Wp // 2)
        g4 = ops.single(N, (Hp // 2 - 1) // 2 + 1, (Wp // 2 - 1) // 2 + 1)
        pl.g_stem, pl.g_pool = g2, g4
        # the fused stem + max-pool launch never materialises the half-resolution stem output (MODEL.FUSE_STEM_POOL False: two launches)
        pl.stem_out, pl.pool_out = (None if self.fuse_stem_pool else act(g2, 64)), act(g4, 64)
        # backbone
        pl.blk = []
        gin = g4
        for blk in self.blocks:
            b = _Plan()
            b.gin = gin
            b.gout = gin.conv_out(1, blk["stride"], 0) if blk["stride"] == 2 else gin
            b.fused = (self.fuse_frozen_blocks and not blk["trainable"] and blk["kind"] == "bottleneck" and blk["stride"] == 1
                       and gin.nlev == 1
                       and ops.bottleneck_fwd_supported(N, gin.H[0], gin.W[0], blk["cin"], blk["ch"], blk["cout"], blk["has_ds"]))
            if b.fused:                                   # one launch: no mid tensors, no materialised shortcut
                b.mid_geo = [gin, b.gout]
                b.mids = []
            elif blk["kind"] == "bottleneck":
                b.mid_geo = [gin, b.gout]
                b.mids = [act(gin, blk["ch"]), act(b.gout, blk["ch"])]
            else:
                b.mid_geo = [b.gout]
                b.mids = [act(b.gout, blk["ch"])]
            b.idt = act(b.gout, blk["cout"]) if (blk["has_ds"] and not b.fused) else None
            b.out = act(b.gout, blk["cout"])
            # e4m3 twin of conv1's output, written by the dense 1x1 launch for the fp8 conv2 that follows
            b.mid8 = None
            if (blk["kind"] == "bottleneck" and blk["convs"][1].fp8
                    and ops.dense_1x1_bits_ok(blk["convs"][0].desc(gin, gin))):
                b.mid8 = torch.empty((gin.pixels, blk["ch"]), dtype=torch.uint8, device=dev)
            b.out_bits = None
            b.g_mid8 = None
            # one-byte twins for the fp8 1x1 launches: conv2's output (conv3 reads it), the block output (the next block's conv1 reads
            # it), and in the backward pass the block's output gradient (conv3's data gradient) and conv2's data gradient (conv1's)
            b.mid8b = b.out8 = b.g_out8 = b.g_mid8a = None
            b.g_out8_ready = False
            if blk["kind"] == "bottleneck":
                u8 = lambda geo, ch: torch.empty((geo.pixels, ch), dtype=torch.uint8, device=dev)
                nxt = self.blocks[len(pl.blk) + 1] if len(pl.blk) + 1 < len(self.blocks) else None
                if blk["convs"][2].fp8_1x1 and blk["convs"][1].fp8 and blk["convs"][1].stride == 1:
                    b.mid8b = u8(b.gout, blk["ch"])
                if nxt is not None and nxt["kind"] == "bottleneck" and nxt["convs"][0].fp8_1x1 \
                        and ops.dense_1x1_bits_ok(blk["convs"][2].desc(b.gout, b.gout)):
                    b.out8 = u8(b.gout, blk["cout"])
                if blk["trainable"] and self.fp8_grad_twins:
                    if blk["convs"][2].fp8_1x1_dgrad and nxt is not None and nxt["kind"] == "bottleneck" and nxt["trainable"]:
                        b.g_out8 = u8(b.gout, blk["cout"])
                    if blk["convs"][0].fp8_1x1_dgrad and blk["convs"][1].fp8_dgrad:
                        b.g_mid8a = u8(gin, blk["ch"])
            if blk["trainable"]:
                b.g_mids = [torch.empty_like(t) for t in b.mids]
                b.g_out = torch.empty_like(b.out)
                if (blk["kind"] == "bottleneck" and blk["convs"][1].fp8_dgrad and self.fp8_grad_twins
                        and blk["convs"][2].dgrad_writes_twin(b.gout, b.gout)):
                    b.g_mid8 = torch.empty((b.gout.pixels, blk["ch"]), dtype=torch.uint8, device=dev)
                # the block output's ReLU gate, bit-packed by the conv3 launch that writes it (1 bit instead of a bf16 per element):
                # what the NEXT block's conv1 / the FPN lateral read as their data-gradient mask (dense 1x1 launches: conv1x1.hip)
                if (blk["kind"] == "bottleneck" and self.use_mask_bits
                        and ops.dense_1x1_bits_ok(blk["convs"][-1].desc(b.gout, b.gout))):
                    b.out_bits = torch.empty((blk["cout"] // 32, b.gout.pixels), dtype=torch.int32, device=dev)
            pl.blk.append(b)
            gin = b.gout
        # feature taps (last block of layer 2..4 -> res3..res5)
        pl.res = {}
        for i, blk in enumerate(self.blocks):
            last = i + 1 == len(self.blocks) or self.blocks[i + 1]["layer"] != blk["layer"]
            if last:
                pl.res[blk["layer"] + 1] = i
        # pyramid
        sizes = [(pl.blk[pl.res[s]].gout.H[0], pl.blk[pl.res[s]].gout.W[0]) for s in self.fpn_stages]
        h5, w5 = sizes[-1]
        h6, w6 = (h5 - 1) // 2 + 1, (w5 - 1) // 2 + 1
        h7, w7 = (h6 - 1) // 2 + 1, (w6 - 1) // 2 + 1
        sizes = sizes + ([(h6, w6), (h7, w7)] if self.TOP_BLOCK == "p6p7" else [(h6, w6)])
        pl.sizes = sizes
        pl.pyr = Geom(N, [s[0] for s in sizes], [s[1] for s in sizes])
        ch = self.fpn_ch
        pl.P = act(pl.pyr, ch)
        pl.g_P = act(pl.pyr, ch)
        # e4m3 twin of the pyramid: written by the fp8 launches that produce P (FPN output convolutions, P6, P7), read by the heads'
        # first convolutions instead of a cast pass.  Only when EVERY level is written by an fp8 convolution (LastLevelP6P7).
        writers = [self.output[s] for s in self.fpn_stages] + ([self.p6, self.p7] if self.TOP_BLOCK == "p6p7" else [])
        pl.P8 = (torch.empty((pl.pyr.pixels, ch), dtype=torch.uint8, device=dev)
                 if self.TOP_BLOCK == "p6p7" and all(c.fp8 for c in writers) else None)
        pl.lat = {s: act(pl.blk[pl.res[s]].gout, ch) for s in self.fpn_stages}
        pl.g_lat = {s: torch.empty_like(pl.lat[s]) for s in self.fpn_stages}
        g6 = pl.pyr.level(len(self.fpn_stages))
        pl.p6_relu = torch.empty((N * h6 * w6, ch), **bf)
        pl.g_p6r = ops.single(N, h6, w6)
        self._plan_head(pl)
        # workspaces
        need = 0
        for blk, b in zip(self.blocks, pl.blk):
            if not blk["trainable"]:
                continue
            geos = [b.gin] + b.mid_geo + [b.gout]
            for ci, c in enumerate(blk["convs"]):
                need = max(need, c.wgrad_ws_bytes(geos[ci], geos[ci + 1]))
            if blk["ds"] is not None:
                need = max(need, blk["ds"].wgrad_ws_bytes(b.gin, b.gout))
        for s in self.fpn_stages:
            gl = pl.blk[pl.res[s]].gout
            need = max(need, self.lateral[s].wgrad_ws_bytes(gl, gl), self.output[s].wgrad_ws_bytes(gl, gl))
        g5 = pl.blk[pl.res[self.fpn_stages[-1]]].gout
        if self.TOP_BLOCK == "p6p7":
            need = max(need, self.p6.wgrad_ws_bytes(g5, g6), self.p7.wgrad_ws_bytes(pl.g_p6r, pl.pyr.level(len(self.fpn_stages) + 1)))
        need = max(need, self._head_wgrad_ws_bytes(pl))
        pl.wgrad_ws = torch.empty((need // 4 + 64,), dtype=torch.float32, device=dev)
        pl.colsum_ws = torch.empty((ops.colsum_workspace_bytes(2048) // 4,), dtype=torch.float32, device=dev)
        self._plans[key] = pl
        return pl

    def _head_wgrad_ws_bytes(self, pl):
        return max(c.wgrad_ws_bytes(pl.pyr, pl.pyr) for c in self._head_convs())

    # ------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------
    def pre_process(self, inputs):
        """RetinaNet.pre_process (retinanet.py:90-107): H2D copy + pad to x32 + normalise (fused kernel)."""
        image = inputs["data"] if isinstance(inputs, dict) else inputs
        if not (torch.is_tensor(image) and image.is_cuda):
            image = self._host_to_device(image)
        image = image.to(self.device, dtype=torch.float32, non_blocking=True).contiguous()
        N, _, H, W = image.shape
        Hp, Wp = _round_up(H, 32), _round_up(W, 32)
        pl = self._plan(N, Hp, Wp)
        ops.pad_normalize(image, Hp, Wp, self.img_mean, self.img_std, pl.x_halo)
        out = {"plan": pl}
        if isinstance(inputs, dict) and "gt_boxes" in inputs:
            gt = torch.as_tensor(np.asarray(inputs["gt_boxes"]), dtype=torch.float32) if not torch.is_tensor(inputs["gt_boxes"]) else inputs["gt_boxes"]
            gt = gt.to(self.device, dtype=torch.float32)
            if gt.shape[1] == 0:
                # a batch without a single annotation (the pad collator then yields (N, 0, 5)): one all-zero padding row keeps the
                # assignment kernels' Gmax > 0 contract; num_gt (im_info[:, 4]) is 0, so every anchor / point is background
                gt = torch.zeros((gt.shape[0], 1, 5), dtype=torch.float32, device=self.device)
            out["gt_boxes"] = gt.contiguous()
        if isinstance(inputs, dict) and "im_info" in inputs:
            info = torch.as_tensor(np.asarray(inputs["im_info"]), dtype=torch.float32) if not torch.is_tensor(inputs["im_info"]) else inputs["im_info"]
        else:
            info = torch.tensor([[Hp, Wp, H, W, 0]] * N, dtype=torch.float32)
        out["img_info"] = info.to(self.device, dtype=torch.float32).contiguous()
        return out

    def _host_to_device(self, image):
        """data_to_input's `Tensor(image)` (layers/common/pre_processing.py:13): a host batch (the loaders yield float64 / float32 /
        uint8 numpy arrays) becomes the fp32 device tensor through bd_h2d_submit -- the library's worker threads convert it chunk by
        chunk into pinned memory and every chunk leaves with its own DMA as soon as it is converted (conversion under transfer).
        Round 2 converted the whole batch with one host copy first: 13-20 ms of the reference-protocol step."""
        arr = image.numpy() if torch.is_tensor(image) else np.asarray(image)
        st = getattr(self, "_stager", None)
        if st is None:
            st = self._stager = ops.HostStager(self.device, int(self.cfg.MODEL.get("H2D_THREADS", 0)))
        if not st.supports(arr):
            arr = np.ascontiguousarray(arr, dtype=np.float32)
        dst = getattr(self, "_h2d_dst", None)
        if dst is None or dst.shape != arr.shape:
            dst = self._h2d_dst = torch.empty(arr.shape, dtype=torch.float32, device=self.device)
        return st.submit(arr, dst, int(self.cfg.MODEL.get("H2D_CHUNK_ELEMS", 0)))

    def _block_forward(self, blk, b, x, x8=None):
        """x8: the e4m3 twin of the block input when the previous block's conv3 wrote one (fp8 mode)."""
        convs = blk["convs"]
        if getattr(b, "fused", False):
            ds = blk["ds"]
            return ops.bottleneck_fwd(b.gin.N, b.gin.H[0], b.gin.W[0], blk["cin"], blk["ch"], blk["cout"], x, convs[0].w_fwd, convs[0].b,
                                      convs[1].w_fwd, convs[1].b, convs[2].w_fwd, convs[2].b, ds.w_fwd if ds is not None else None,
                                      ds.b if ds is not None else None, b.out)
        idt = x
        if blk["ds"] is not None:
            blk["ds"].forward(x, b.gin, b.gout, b.idt)
            idt = b.idt
        geos = [b.gin] + b.mid_geo + [b.gout]
        t, t8 = x, x8
        for ci, c in enumerate(convs[:-1]):
            y8 = None
            if ci == 0 and getattr(b, "mid8", None) is not None:
                y8 = b.mid8
            elif ci == 1 and getattr(b, "mid8b", None) is not None:
                y8 = b.mid8b
            c.forward(t, geos[ci], geos[ci + 1], b.mids[ci], relu=True, x8=t8, y8=y8,
                      q_scale=convs[ci + 1].act_scale if y8 is not None else 1.0)
            t, t8 = b.mids[ci], y8
        convs[-1].forward(t, geos[-2], geos[-1], b.out, add=idt, relu=True, bits=b.out_bits, x8=t8, y8=getattr(b, "out8", None))
        return b.out

    def network_forward(self, pl):
        """RetinaNet.network_forward (retinanet.py:109-118): backbone + FPN + head; logits/offsets come out already
        in the (N, sum HWA, K) layout of permute_to_N_Any_K + concat (function.py:26-32, retinanet.py:127-132)."""
        N = pl.N
        if self.fuse_stem_pool:
            ops.stem_pool_fwd(N, pl.Hp, pl.Wp, pl.x_halo, self.stem_packed, self.stem_shift, pl.pool_out)
        else:
            ops.stem_conv7x7_fwd(N, pl.Hp, pl.Wp, pl.x_halo, self.stem_packed, self.stem_shift, pl.stem_out)
            ops.maxpool3x3s2_fwd(pl.stem_out, N, pl.g_stem.H[0], pl.g_stem.W[0], 64, pl.pool_out)
        x = pl.pool_out
        x8 = None
        for blk, b in zip(self.blocks, pl.blk):
            x = self._block_forward(blk, b, x, x8)
            x8 = getattr(b, "out8", None)
        # FPN (fpn_backbone.py:123-160): top-down from the coarsest level
        st = self.fpn_stages
        nl = len(st)
        b5 = pl.blk[pl.res[st[-1]]]
        side = self._wstream if (self.TOP_BLOCK == "p6p7" and self.async_wgrad and self._wstream is not None) else None
        if self.TOP_BLOCK == "p6p7":
            # LastLevelP6P7 (:198-204) only needs res5 and writes its own pyramid levels: its two small-grid convs (70 workgroups for
            # P6 at 800x1344) run on the side stream, concurrently with the lateral / output convs below
            g6, g7 = pl.pyr.level(nl), pl.pyr.level(nl + 1)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side) if side is not None else _nullcontext():
                self.p6.forward(b5.out, b5.gout, g6, pl.P, y8=pl.P8)
                self._relu_level(pl.P, g6, pl.p6_relu)
                self.p7.forward(pl.p6_relu, pl.g_p6r, g7, pl.P, y8=pl.P8)
        prev, prev_geo = None, None
        for li in range(nl - 1, -1, -1):
            s = st[li]
            b = pl.blk[pl.res[s]]
            self.lateral[s].forward(b.out, b.gout, b.gout, pl.lat[s])
            if prev is not None:
                ops.upsample2x_add_fwd(prev, prev_geo, pl.lat[s], b.gout, self.fpn_ch)
            self.output[s].forward(pl.lat[s], b.gout, pl.pyr.level(li), pl.P, y8=pl.P8)
            prev, prev_geo = pl.lat[s], b.gout
        if self.TOP_BLOCK == "p6p7":
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
        else:
            ops.subsample2x_fwd(pl.P, pl.pyr.level(nl - 1), pl.P, pl.pyr.level(nl), self.fpn_ch)   # FPNP6 (:172-183)
        self.head_forward(pl)

    def _relu_level(self, src, geo, dst):
        """dst (dense, per level) = relu(one pyramid level of src); the level is contiguous per image."""
        C = src.shape[1]
        n = geo.H[0] * geo.W[0]
        sv = src.view(geo.N, geo.pix_per_img, C)
        dv = dst.view(geo.N, n, C)
        for i in range(geo.N):
            ops.relu_bf16(sv[i, geo.off[0]: geo.off[0] + n], dv[i])

    # ------------------------------------------------------------------------------------------------
    # inference post-processing shared by the heads (layers/common/post_processing.py:50-103)
    # ------------------------------------------------------------------------------------------------
    def _detect(self, scores, lvl_rows, K, mode, info, k=1000, anchors=None, offsets=None, off_ld=4, A=1, mean=(0, 0, 0, 0),
                std=(1, 1, 1, 1), item_boxes=None):
        """scores: fp32 [sum(lvl_rows) * K] on the device.  Per level: score > TEST.CLS_THRESHOLD -> top-k (descending) ->
        label = idx % K, box of row idx // K; then batched NMS by label, keep MAX_BOXES_PER_IMAGE, rescale + clip.
        Everything stays on the device; the only host read is the final detection count."""
        from ..structures import Boxes, Container
        t = self.cfg.TEST
        dev = self.device
        Ln = len(lvl_rows)
        row_off = [0]
        for r in lvl_rows[:-1]:
            row_off.append(row_off[-1] + r)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        tk_idx = torch.empty((Ln, k), **i32); tk_sc = torch.empty((Ln, k), **f32); tk_cnt = torch.empty((Ln,), **i32)
        ops.segment_topk(scores, 1, 0, 1, 1, 0, [r * K for r in row_off], [r * K for r in lvl_rows], k, tk_idx, tk_sc, tk_cnt,
                         min_score=t.CLS_THRESHOLD)
        C = Ln * k
        boxes = torch.empty((C, 4), **f32); sc = torch.empty((1, C), **f32); labels = torch.empty((1, C), **i32)
        ops.det_candidates(mode, tk_idx, tk_sc, tk_cnt, Ln, k, row_off, K, anchors, offsets, off_ld, A, mean, std, item_boxes,
                           boxes, sc, labels)
        max_out = t.MAX_BOXES_PER_IMAGE
        keep = torch.empty((1, max_out), **i32); num = torch.zeros((1,), **i32)
        ws = torch.empty((ops.nms_batched_workspace_bytes(1, C),), dtype=torch.uint8, device=dev)
        ops.nms_batched(boxes, sc, labels, t.IOU_THRESHOLD, max_out, keep, num, ws)
        ob = torch.empty((max_out, 4), **f32); osc = torch.empty((max_out,), **f32); ol = torch.empty((max_out,), **i32)
        ops.det_finalize(boxes, sc, labels, keep, num, max_out, info[0].contiguous(), ob, osc, ol)
        n = int(num.item())
        if n == 0:
            e = torch.zeros((0,))
            return Container(boxes=e, box_scores=e, box_labels=e)
        return Container(boxes=Boxes(ob[:n]), box_scores=osc[:n], box_labels=ol[:n])

    # ------------------------------------------------------------------------------------------------
    # backward (replaces GradManager.backward, solver/default_solver.py:118-124)
    # ------------------------------------------------------------------------------------------------
    def _wgrad(self, conv, x, g, gin, gout, ws, cws=None, x8=None, g8=None):
        """conv.wgrad on the side stream: it only needs x and g as they are NOW (everything enqueued so far on the main
        stream), and nothing on the main stream reads its outputs before `_join_wgrads`.  Callers must not overwrite g/x
        later in the same backward pass (the heads keep one gradient buffer per layer for that reason)."""
        q = None
        if self.wgrad_queue_mode != "layer" and self.device.type == "cuda" and not (conv.fp8_wgrad and x8 is not None and g8 is not None):
            # deferred reduce: this layer's partial sums get their own slice of the model's arena, untouched until the next flush (a
            # gradient bucket's end, or the byte threshold).  A layer that does not fit (the first backward pass of a model, a larger
            # input size, another set of queued layers) runs un-queued on the plan's shared workspace -- same bits -- and the arena is
            # re-grown to the recorded peak at _join_wgrads
            key = (conv.name, gin.N, tuple(gin.H), tuple(gin.W), tuple(gout.H), tuple(gout.W))
            need = self._wq_need.get(key)
            if need is None:
                need = self._wq_need[key] = (conv.wgrad_ws_bytes(gin, gout) + 255) // 256 * 256
            off = self._wq_off
            self._wq_off = off + need
            arena = self._wq_arena
            if arena is not None and off + need <= arena.numel() * 4:
                ws = arena[off // 4: (off + need) // 4]
                self._wq_pending += need
                if self._wq is None:
                    self._wq = ops.WgradQueue()
                q = self._wq
        if not (self.async_wgrad and self._wstream is not None):
            conv.wgrad(x, g, gin, gout, ws, cws, x8=x8, g8=g8, queue=q)
        else:
            self._wstream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._wstream):
                conv.wgrad(x, g, gin, gout, ws, cws, x8=x8, g8=g8, queue=q)
        if q is not None and isinstance(self.wgrad_queue_mode, int) and self._wq_pending >= self.wgrad_queue_mode:
            self._flush_wgrads()

    def _begin_wgrads(self):
        """Start of a backward pass (also the head modules' own, layers/modules.py): the arena is free from offset 0 -- UNLESS partial sums
        are already waiting for the head bucket's flush: Faster R-CNN runs its RPN head's backward inside get_losses, under the proposal
        chain, and those slices must survive until then.  (Rounds 4's reset here let the box head's kernels overwrite them: the RPN
        weight gradients of every queued step were wrong -- found by tests/test_wgrad_queue_gpu.py, the first test of the queued path.)"""
        if self._wq is None or not self._wq.pending():
            self._wq_off = self._wq_pending = 0

    def _flush_wgrads(self):
        """One launch reduces every weight gradient queued since the last flush (on the stream the partial sums were computed on); the
        arena is free again from offset 0 for the kernels enqueued behind that reduce."""
        self._wq_peak = max(self._wq_peak, self._wq_off)
        self._wq_off = self._wq_pending = 0
        q = self._wq
        if q is None or not q.pending():
            return
        if self.async_wgrad and self._wstream is not None:
            with torch.cuda.stream(self._wstream):
                q.flush()
        else:
            q.flush()

    def _join_wgrads(self):
        self._flush_wgrads()
        if self.async_wgrad and self._wstream is not None:
            torch.cuda.current_stream().wait_stream(self._wstream)
        have = 0 if self._wq_arena is None else self._wq_arena.numel() * 4
        if self.wgrad_queue_mode != "layer" and self._wq_peak > have:
            # (behind the join: the old arena's last readers have been ordered in front of the current stream, which owns both allocations)
            self._wq_arena = None
            self._wq_arena = torch.empty((self._wq_peak // 4 + 64,), dtype=torch.float32, device=self.device)

    def backward(self, on_bucket_ready=None):
        pl = self._cur
        probing = self._fp8_probe_begin()
        self._begin_wgrads()
        ws, cws = pl.wgrad_ws, pl.colsum_ws
        pyr = pl.pyr
        pl.g_P8_ready = False                       # set by a head whose last data gradients wrote the e5m2 twin of dL/dP
        for b in pl.blk:
            b.g_out8_ready = False
        self.head_backward(pl, ws, cws)
        side = (self._wstream,) if (self.async_wgrad and self._wstream is not None) else ()
        self._flush_wgrads()
        if on_bucket_ready:
            on_bucket_ready("head", side)
        # ---- FPN
        st = self.fpn_stages
        nl = len(st)
        b5 = pl.blk[pl.res[st[-1]]]
        pool_top = self.TOP_BLOCK != "p6p7"
        if not pool_top:
            g6, g7 = pyr.level(nl), pyr.level(nl + 1)
            # P7 = conv(relu(P6)): d P6 = dgrad(g_P7) * (P6 > 0) + g_P6(head), written in place into g_P's P6 level
            # The two dgrads are small grids that only touch the P6/P7 levels of g_P and res5's gradient, which the main stream
            # does not read before the top lateral dgrad below: they run on their own stream next to the P3.. output-conv dgrads.
            top = self._tstream if self.async_wgrad else None
            self._wgrad(self.p7, pl.p6_relu, pl.g_P, pl.g_p6r, g7, ws, cws)
            if top is not None:
                top.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(top) if top is not None else _nullcontext():
                self.p7.dgrad(pl.g_P, g6, g7, pl.g_P, mask=pl.P, add_after=pl.g_P)
                self._wgrad(self.p6, b5.out, pl.g_P, b5.gout, g6, ws, cws)
                self.p6.dgrad(pl.g_P, b5.gout, g6, b5.g_out, first=True)
        else:
            ops.subsample2x_bwd_add(pl.g_P, pyr.level(nl), pl.g_P, pyr.level(nl - 1), self.fpn_ch)      # P6 = P5[::2, ::2]
        for li in range(nl):
            s = st[li]
            b = pl.blk[pl.res[s]]
            lvl = pyr.level(li)
            self._wgrad(self.output[s], pl.lat[s], pl.g_P, b.gout, lvl, ws, cws)
            self.output[s].dgrad(pl.g_P, b.gout, lvl, pl.g_lat[s], first=True, g8=pl.g_P8 if pl.g_P8_ready else None)
            if li > 0:   # gradient arriving through the top-down path from the finer level
                sf = st[li - 1]
                ops.upsample2x_add_bwd(pl.g_lat[sf], pl.blk[pl.res[sf]].gout, pl.g_lat[s], b.gout, self.fpn_ch, accumulate=True)
            self._wgrad(self.lateral[s], b.out, pl.g_lat[s], b.gout, b.gout, ws, cws)
            # res_s gradient: first contribution for res3/res4, second (after P6) and final for res5 -> mask there
            is_top = li == nl - 1
            if not self.blocks[pl.res[s]]["trainable"]:
                continue                                   # res2 of a FREEZE_AT=2 backbone: nothing below needs the gradient
            if is_top:
                if not pool_top and self._tstream is not None and self.async_wgrad:
                    torch.cuda.current_stream().wait_stream(self._tstream)
                self.lateral[s].dgrad(pl.g_lat[s], b.gout, b.gout, b.g_out, first=pool_top, mask=b.out, maskbits=b.out_bits)
            else:
                self.lateral[s].dgrad(pl.g_lat[s], b.gout, b.gout, b.g_out, first=True)
        self._flush_wgrads()
        if on_bucket_ready:
            on_bucket_ready("fpn", side)
        # ---- backbone, last block first.  g_out of a block holds the masked gradient once all consumers are done:
        # res5: done above.  res3/res4 (and every inner block output): the next block's dgrads finish it.
        nb = len(self.blocks)
        for bi in range(nb - 1, -1, -1):
            blk, b = self.blocks[bi], pl.blk[bi]
            if not blk["trainable"]:
                break
            convs = blk["convs"]
            geos = [b.gin] + b.mid_geo + [b.gout]
            xin = pl.blk[bi - 1].out if bi > 0 else pl.pool_out
            prev_tr = bi > 0 and self.blocks[bi - 1]["trainable"]
            G = b.g_out
            # the shortcut convolution's weight gradient needs only the block's input and output gradient: first in the side
            # stream's queue, not last (after the final block nothing is left on the main stream to hide it)
            if blk["ds"] is not None:
                self._wgrad(blk["ds"], xin, G, b.gin, b.gout, ws)
            # main branch, last conv backwards
            g = G
            # fp8 mode: the e5m2 twin of the block's output gradient, written by the next block's conv1 data gradient (its last writer)
            g8 = b.g_out8 if (getattr(b, "g_out8", None) is not None and b.g_out8_ready) else None
            for ci in range(len(convs) - 1, 0, -1):
                # conv2's weight gradient from the twins both neighbours wrote (conv1's forward output, conv3's data gradient)
                wx8 = getattr(b, "mid8", None) if (ci == 1 and len(convs) == 3) else None
                self._wgrad(convs[ci], b.mids[ci - 1], g, geos[ci], geos[ci + 1], ws, x8=wx8, g8=g8 if wx8 is not None else None)
                # conv3's (dense 1x1) data gradient also writes the e5m2 twin that conv2's fp8 data gradient reads, conv2's the one
                # conv1's reads
                nxt = None
                if len(convs) == 3:
                    nxt = getattr(b, "g_mid8", None) if ci == 2 else getattr(b, "g_mid8a", None)
                if nxt is not None and not convs[ci].dgrad_writes_twin(geos[ci], geos[ci + 1]):
                    nxt = None
                wrote = convs[ci].dgrad(g, geos[ci], geos[ci + 1], b.g_mids[ci - 1], mask=b.mids[ci - 1], g8=g8, dx8=nxt,
                                        q_scale=convs[ci - 1].grad_scale)
                g, g8 = b.g_mids[ci - 1], (nxt if wrote else None)
            self._wgrad(convs[0], xin, g, geos[0], geos[1], ws)
            if prev_tr:
                gx = pl.blk[bi - 1].g_out
                xbits = pl.blk[bi - 1].out_bits if ops.dense_1x1_bits_ok(convs[0].desc(geos[0], geos[1])) else None
                # has the input already received a contribution (FPN lateral of res3/res4)?
                tapped = any(pl.res[s] == bi - 1 for s in st)
                # conv1's data gradient is the LAST writer of the previous block's output gradient: it also writes that gradient's e5m2
                # twin (fp8 mode) for the previous block's conv3
                pb = pl.blk[bi - 1]
                gx8 = getattr(pb, "g_out8", None)
                # ... written with the scale of its consumer, the previous block's conv3 (another scale group at a layer boundary)
                kw = dict(mask=xin, maskbits=xbits, g8=g8, dx8=gx8, q_scale=self.blocks[bi - 1]["convs"][-1].grad_scale)
                if blk["ds"] is not None and gx8 is None and self.sparse_shortcut_grad:
                    # conv1 first (writes every pixel, gated), then the stride-2 shortcut adds its gradient IN PLACE at the quarter of
                    # the pixels it reaches ((a m + b) m = (a + b) m for a 0 / 1 gate m): instead of a full-resolution tensor that is
                    # three quarters zeros being written, read back and summed
                    wrote = convs[0].dgrad(g, geos[0], geos[1], gx, first=not tapped, **kw)
                    blk["ds"].dgrad(G, b.gin, b.gout, gx, first=False, mask=xin, sparse=True)
                elif blk["ds"] is not None:         # (the last writer must be the launch that can write the e5m2 twin)
                    blk["ds"].dgrad(G, b.gin, b.gout, gx, first=not tapped)
                    wrote = convs[0].dgrad(g, geos[0], geos[1], gx, first=False, **kw)
                else:
                    if tapped:
                        ops.add_bf16(gx, G, gx)
                        wrote = convs[0].dgrad(g, geos[0], geos[1], gx, first=False, **kw)
                    else:
                        # identity skip: gx = (dgrad + G) * mask
                        wrote = convs[0].dgrad(g, geos[0], geos[1], gx, add_before=G, **kw)
                pb.g_out8_ready = gx8 is not None and bool(wrote)     # what the launch actually did (a bf16 fallback drops the twin)
            if bi == 0 or self.blocks[bi - 1]["layer"] != blk["layer"]:
                self._flush_wgrads()
                if on_bucket_ready:
                    on_bucket_ready(f"layer{blk['layer']}", side)
        self._join_wgrads()
        self._fp8_probe_end(probing)

