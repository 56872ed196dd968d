"""Faster R-CNN (basedet/models/det/faster_rcnn.py, rpn.py, layers/head/rcnn.py) on the HIP path: ResNet-FPN trunk with
P2-P6 (FPNP6), RPN head + proposal selection inside the training step, RoI sampling, multi-level RoIAlign and the
two-FC box head (the FC layers run on the 1x1 implicit-GEMM kernels, RoIs as "pixels").

``model(batch)`` returns ``{"total_loss", "rpn_cls_loss", "rpn_reg_loss", "rcnn_cls_loss", "rcnn_reg_loss"}``
(faster_rcnn.py:84-96).  Static shapes: every image owns RPN.TRAIN_POST_NMS_TOPK proposal slots and RCNN.NUM_ROIS sample
slots (label -1 = empty slot, zero loss / zero gradient); nothing is synchronised with the host inside the step.

Randomness: the reference subsamples anchors and RoIs with megengine.random.uniform keys (layers/common/sampling.py:26).
Here the keys come from a seeded device generator, or from ``batch["sample_keys"]`` (tests feed the same keys to the oracle).
"""
import math

import numpy as np
import torch

from .. import ops
from ..utils.registry import registers
from . import params as P
from .engine import FCLayer, FusedPredConv
from .fpn_base import FPNDetector, _round_up


@registers.models.register()
class FasterRCNN(FPNDetector):
    TOP_BLOCK = "pool"

    @staticmethod
    def init_params(cfg, seed=0):
        return P.init_faster_rcnn_params(cfg, seed)

    # ---- construction ------------------------------------------------------------------------------------
    def _build_head(self, add, params):
        m = self.cfg.MODEL
        dev = self.device
        ch = self.fpn_ch
        A = self.num_anchors = len(m.ANCHOR.SCALES[0]) * len(m.ANCHOR.RATIOS[0])
        rc = m.RPN.CHANNELS
        # RPN (rpn.py:52-62): 3x3 conv + ReLU, then objectness (A) and offsets (4A) 1x1 convs fused into one launch
        self.rpn_conv = add("rpn.rpn_conv", ch, rc, 3, 1, 1, bias=True)
        self.rpn_ld = _round_up(5 * A, 8)
        self.rpn_pred = FusedPredConv("rpn.pred", [("rpn.rpn_cls_score", A), ("rpn.rpn_bbox_offsets", 4 * A)], rc, 1, 1, 0, dev,
                                      cout_pad=self.rpn_ld)
        self.convs[self.rpn_pred.name] = self.rpn_pred
        # RCNN (layers/head/rcnn.py:32-38)
        self.pool = tuple(m.ROI_POOLER.SIZE)
        assert m.ROI_POOLER.METHOD == "roi_align", "the HIP path implements roi_align"
        self.rcnn_levels = len(m.RCNN.IN_FEATURES)
        assert list(m.RCNN.STRIDES) == self.strides[: self.rcnn_levels]
        K = self.num_classes
        fin = ch * self.pool[0] * self.pool[1]
        self.fc1 = FCLayer("rcnn.fc1", fin, 1024, dev, in_chw=(ch, self.pool[0], self.pool[1]))
        self.fc2 = FCLayer("rcnn.fc2", 1024, 1024, dev)
        self.rcnn_ld = _round_up(K + 1 + 4 * K, 8)
        self.rcnn_pred = FCLayer("rcnn.pred", 1024, K + 1 + 4 * K, dev, parts=[("rcnn.pred_cls", K + 1), ("rcnn.pred_delta", 4 * K)],
                                 cout_pad=self.rcnn_ld)
        for c in (self.fc1, self.fc2, self.rcnn_pred):
            self.convs[c.name] = c
        scales = np.asarray(m.ANCHOR.SCALES, np.float32).tolist()
        ratios = np.asarray(m.ANCHOR.RATIOS, np.float32).tolist()
        if len(ratios) == 1:
            ratios = ratios * len(self.strides)
        if len(scales) == 1:
            scales = scales * len(self.strides)
        self.base_anchors = []
        for sc_, ra_ in zip(scales, ratios):     # layers/common/anchor_generator.py:95-109
            base = []
            for s_ in sc_:
                area = float(s_) ** 2.0
                for r_ in ra_:
                    w = math.sqrt(area / float(r_)); h = float(r_) * w
                    base.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
            self.base_anchors.append(torch.tensor(base, dtype=torch.float32, device=dev))
        self.pre_k = {True: m.RPN.TRAIN_PREV_NMS_TOPK, False: m.RPN.TEST_PREV_NMS_TOPK}
        self.post_k = {True: m.RPN.TRAIN_POST_NMS_TOPK, False: m.RPN.TEST_POST_NMS_TOPK}
        # RoIAlign backward: the tiled fixed-order sum (bd_roi_align_bwd_bf16: per-tile RoI lists in slot order, sums in registers, written
        # once on top of the RPN head's dL/dP; bitwise reproducible) wherever its kernel's shape limits hold -- the configured 7 x 7 pooler,
        # <= 512 RoIs per image, an even channel count; any other ROI_POOLER.SIZE / NUM_ROIS (the reference accepts them: roi_pool.py:35-78)
        # takes the general fp32 scatter + conversion pass (bd_roi_align_bwd, the default of rounds 2-4).  deterministic_roi_bwd = False
        # forces the scatter (tests: both forms against the oracle).
        self.deterministic_roi_bwd = True
        self.thin_rpn_bwd = bool(m.get("THIN_RPN_BWD", True))     # RPN prediction layer: fused one-pass backward (0: the generic weight / data gradient kernels)
        self._gen = torch.Generator(device=dev) if self.device.type == "cuda" else None
        if self._gen is not None:
            self._gen.manual_seed(0)

    def _head_convs(self):
        return [self.rpn_conv, self.rpn_pred, self.fc1, self.fc2, self.rcnn_pred]

    def _head_wgrad_ws_bytes(self, pl):
        need = max(self.rpn_conv.wgrad_ws_bytes(pl.pyr, pl.pyr), self.rpn_pred.wgrad_ws_bytes(pl.pyr, pl.pyr))
        return max(need, *(c.wgrad_ws_bytes(pl.g_fc, pl.g_fc) for c in (self.fc1, self.fc2, self.rcnn_pred)))

    def _plan_head(self, pl):
        dev = self.device
        m = self.cfg.MODEL
        N = pl.N
        A = self.num_anchors
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        pyr = pl.pyr
        rc = m.RPN.CHANNELS
        pl.rpn_t = torch.empty((pyr.pixels, rc), **bf)
        pl.rpn_raw = torch.empty((pyr.pixels, self.rpn_ld), **bf)
        pl.d_rpn_raw = torch.zeros((pyr.pixels, self.rpn_ld), **bf)        # padding channel stays zero
        pl.g_rpn_t = torch.empty((pyr.pixels, rc), **bf)
        tot = pyr.pix_per_img * A
        pl.A_total = tot
        pl.anchors = torch.empty((tot, 4), **f32)
        o = 0
        for (h, w), s, base in zip(pl.sizes, self.strides, self.base_anchors):
            n = h * w * A
            ops.anchors_generate(h, w, s, m.ANCHOR.OFFSET, base, pl.anchors[o:o + n])
            o += n
        pl.rpn_labels = torch.empty((N, tot), **i32)
        pl.rpn_match = torch.empty((N, tot), **i32)
        pl.rpn_offsets = torch.empty((N, tot, 4), **f32)
        pl.rpn_num_fg = torch.zeros((1,), **i32)
        pl.rpn_num_valid = torch.zeros((1,), **i32)
        pl.assign_ws = None                   # scratch of bd_rpn_assign_encode (N x Gmax floats): sized by the first batch
        lvl_pixels = [h * w for h, w in pl.sizes]
        post = self.post_k[True]
        assert self.post_k[False] == post, "train / test post-NMS top-k share the proposal slots"
        pl.rois = torch.empty((N, post, 4), **f32)
        pl.num_rois = torch.zeros((N,), **i32)
        pl.prop_ws = torch.empty((max(ops.rpn_proposals_workspace_bytes(N, lvl_pixels, A, k, post) for k in self.pre_k.values()),),
                                 dtype=torch.uint8, device=dev)
        S = m.RCNN.NUM_ROIS
        R = N * S
        pl.R = R
        pl.s_rois = torch.empty((N, S, 4), **f32)
        pl.s_labels = torch.empty((N, S), **i32)
        pl.s_targets = torch.empty((N, S, 4), **f32)
        pl.s_count = torch.zeros((N,), **i32)
        pl.s_total = torch.zeros((1,), **i32)
        ch = self.fpn_ch
        fin = ch * self.pool[0] * self.pool[1]
        pl.pooled = torch.empty((R, fin), **bf)
        pl.fc1_out = torch.empty((R, 1024), **bf)
        pl.fc2_out = torch.empty((R, 1024), **bf)
        pl.rcnn_raw = torch.empty((R, self.rcnn_ld), **bf)
        pl.d_rcnn_raw = torch.empty((R, self.rcnn_ld), **bf)
        pl.g_fc2 = torch.empty((R, 1024), **bf)
        pl.g_fc1 = torch.empty((R, 1024), **bf)
        pl.g_pooled = torch.empty((R, fin), **bf)
        pl.g_feat32 = None                # (the fp32 scatter's staging pyramid, 1.5 GB at batch 16: allocated when that variant runs)
        pl.roi_bwd_tiled = self.pool == (7, 7) and S <= 512 and self.fpn_ch % 2 == 0       # the tiled kernel's limits (rcnn_ops.hip)
        pl.roi_bwd_ws = (torch.empty((ops.roi_align_bwd_bf16_workspace_bytes(pyr, S),), dtype=torch.uint8, device=dev)
                         if pl.roi_bwd_tiled else None)
        pl.g_fc = ops.single(1, R, 1)
        pl.loss_buf = torch.zeros((4,), **f32)

    # ---- forward -----------------------------------------------------------------------------------------
    def head_forward(self, pl):
        """RPN.forward predictions (rpn.py:78-100), all five levels per launch."""
        self.rpn_conv.forward(pl.P, pl.pyr, pl.pyr, pl.rpn_t, relu=True)
        if self.thin_rpn_bwd and self.rpn_pred.thin_backward_ok(pl.pyr):        # (the prediction layer on its own kernels: csrc/conv1x1_thin.hip)
            self.rpn_pred.thin_forward(pl.rpn_t, pl.pyr, pl.rpn_raw)
        else:
            self.rpn_pred.forward(pl.rpn_t, pl.pyr, pl.pyr, pl.rpn_raw)

    def _keys(self, inputs, name, shape):
        sk = inputs.get("sample_keys") if isinstance(inputs, dict) else None
        if sk is not None and name in sk:
            k = sk[name]
            k = torch.as_tensor(np.asarray(k), dtype=torch.float32) if not torch.is_tensor(k) else k
            k = k.to(self.device, dtype=torch.float32).contiguous()
            assert tuple(k.shape) == tuple(shape), f"sample_keys[{name}] has shape {tuple(k.shape)}, expected {tuple(shape)}"
            return k
        return torch.rand(shape, generator=self._gen, device=self.device, dtype=torch.float32)

    def _proposals(self, pl, img_info):
        m = self.cfg.MODEL
        A = self.num_anchors
        ops.rpn_proposals(pl.rpn_raw, self.rpn_ld, A, 0, A, pl.pyr, pl.anchors, img_info, m.RPN_BOX_REG.MEAN, m.RPN_BOX_REG.STD,
                          self.pre_k[self.training], m.RPN.NMS_THRESHOLD, pl.rois.shape[1], pl.rois, pl.num_rois, pl.prop_ws)

    def get_losses(self, inputs):
        """FasterRCNN.get_losses (faster_rcnn.py:78-97) = RPN.forward (rpn.py:70-132) + RCNN.forward (rcnn.py:52-83)."""
        assert self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        self._cur = pl
        m = self.cfg.MODEL
        A = self.num_anchors
        N = pl.N
        gt = pre["gt_boxes"]
        info = pre["img_info"]
        num_gt = info[:, 4].to(torch.int32).contiguous()
        Gmax = gt.shape[1]
        thr = m.MATCHER.THRESHOLDS
        nsa = m.RPN.NUM_SAMPLE_ANCHORS
        S = m.RCNN.NUM_ROIS
        key_ld = pl.rois.shape[1] + Gmax
        # The four random-key tensors of a step (sampling.py:26 draws them where it needs them) are drawn HERE, on the main stream, in one
        # fixed order: which side stream consumes them -- and whether the RPN targets run early -- no longer changes what a seed produces.
        # The plan keeps them until the next step's draw (the side streams that read them have joined by then).
        keys = {name: self._keys(inputs, name, shape) for name, shape in
                (("rpn_pos", (N, pl.A_total)), ("rpn_neg", (N, pl.A_total)), ("rcnn_fg", (N, key_ld)), ("rcnn_bg", (N, key_ld)))}
        pl.sample_keys = keys

        def rpn_targets():
            """RPN.get_ground_truth (rpn.py:215-240): anchors, ground truth and random keys in, labels / offsets out -- nothing of the network."""
            ops.rpn_assign_encode(pl.anchors, gt, num_gt, thr[0], thr[1], m.MATCHER.ALLOW_LOW_QUALITY, m.RPN_BOX_REG.MEAN,
                                  m.RPN_BOX_REG.STD, pl.rpn_labels, pl.rpn_match, pl.rpn_offsets, pl.rpn_num_fg, pl.assign_ws)
            ops.sample_labels(pl.rpn_labels, keys["rpn_pos"], keys["rpn_neg"], int(m.RPN.POSITIVE_ANCHOR_RATIO * nsa), nsa, pl.rpn_num_valid)
            if not (self.deterministic_roi_bwd and pl.roi_bwd_tiled):     # (the fp32 scatter only)
                if pl.g_feat32 is None:
                    pl.g_feat32 = torch.empty((pl.pyr.pixels, self.fpn_ch), dtype=torch.float32, device=self.device)
                pl.g_feat32.zero_()           # the fp32 pyramid RoIAlign's backward scatters into (1.5 GB at batch 16): cleared here, not in backward
                pl.g_feat32_clean = True

        # Round 5: the RPN targets (0.7 ms of one-workgroup-per-image kernels at batch 16: gt_rowmax, assignment, the radix select over
        # 268 569 keys per image) and that clear run on the weight-gradient stream, which is idle during the forward pass, UNDER the
        # backbone -- as RetinaNet's assignment does (rounds 1-4 ran them between the forward and the RPN losses, on the main chain).
        early = self._wstream if (self.async_wgrad and self._wstream is not None and m.get("RPN_TARGETS_EARLY", True)) else None
        if pl.assign_ws is None or pl.assign_ws.numel() < N * Gmax:
            pl.assign_ws = torch.empty((N * Gmax,), dtype=torch.float32, device=self.device)
        if early is not None:
            early.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(early):
                rpn_targets()
        self.network_forward(pl)
        # ---- RPN: proposals (detached), targets, losses.  The proposal chain (decode, per-level top-k, NMS level by level + merge: small
        # grids) runs on a side stream under the RPN losses and the RPN head's backward on the main one.
        # (Round 5, measured and REMOVED: the other way round -- RPN losses + the RPN head's backward on the side stream until the RoIAlign
        # backward's sum into dL/dP, under the whole proposal / box-head chain: 523-524 img/s against 545-547 on one box,
        # profiles/r05_frcnn_ab.txt -- the persistent one-workgroup-per-CU convolution kernels keep the box chain's many small grids
        # waiting for a CU, and the box chain is the critical path; that schedule also failed the bench-batch parity test once.)
        side = self._tstream if (self.async_wgrad and self._tstream is not None) else None

        def sample():
            ops.rcnn_sample_targets(pl.rois, pl.num_rois, gt, num_gt, keys["rcnn_fg"], keys["rcnn_bg"], S, int(S * m.RCNN.FG_RATIO), m.RCNN.FG_THRESHOLD,
                                    m.RCNN.BG_THRESHOLD_HIGH, m.RCNN.BG_THRESHOLD_LOW, m.RCNN_BOX_REG.MEAN, m.RCNN_BOX_REG.STD,
                                    pl.s_rois, pl.s_labels, pl.s_targets, pl.s_count, pl.s_total)

        sample_on_side = bool(m.get("RCNN_SAMPLE_ON_SIDE", True))
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._proposals(pl, info)
                if sample_on_side:        # the RoI sampling (one workgroup per image, 0.23 ms) right behind the proposals, under the RPN head's backward
                    sample()
        else:
            self._proposals(pl, info)
        if early is not None:
            torch.cuda.current_stream().wait_stream(early)
        else:
            rpn_targets()
        pl.loss_buf.zero_()
        ops.rpn_loss_fwd_bwd(pl.rpn_raw, self.rpn_ld, A, 0, A, pl.rpn_labels, pl.rpn_offsets, pl.pyr.pixels,
                             m.LOSSES.RPN_SMOOTH_L1_BETA, pl.rpn_num_valid, pl.loss_buf[0:2], pl.d_rpn_raw)
        pl.rpn_bwd_done = False
        if side is not None:
            # the RPN head's backward needs nothing from the proposal chain: it runs now, under it, and leaves the FIRST contribution
            # to dL/dP; the RoIAlign backward joins it later (head_backward: the tiled sums are added on top, the fp32 scatter's pyramid
            # through an accumulating conversion)
            self._flush_wgrads()          # (partial sums left by a get_losses() that was never followed by backward(): reduce them now, free the arena)
            self._rpn_head_backward(pl, pl.wgrad_ws, pl.colsum_ws, first=True)
            pl.rpn_bwd_done = True
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        # ---- RCNN: sampling, RoIAlign, box head, losses
        if side is None or not sample_on_side:
            sample()
        self._box_head(pl)
        ops.rcnn_loss_fwd_bwd(pl.rcnn_raw, self.rcnn_ld, self.num_classes, self.num_classes + 1, pl.s_labels, pl.s_targets, pl.R,
                              m.LOSSES.RCNN_SMOOTH_L1_BETA, pl.s_total, pl.loss_buf[2:4], pl.d_rcnn_raw)
        lb = pl.loss_buf
        return {"total_loss": lb[0] + lb[1] + lb[2] + lb[3], "rpn_cls_loss": lb[0], "rpn_reg_loss": lb[1],
                "rcnn_cls_loss": lb[2], "rcnn_reg_loss": lb[3]}

    def _box_head(self, pl):
        """roi_pool + fc1/fc2 + predictors (rcnn.py:55-63) on the sampled RoI slots."""
        S = pl.s_rois.shape[1]
        ops.roi_align_fwd(pl.P, pl.pyr, self.rcnn_levels, self.strides, self.fpn_ch, pl.s_rois.view(-1, 4), pl.s_labels.view(-1), S,
                          self.pool, 2, pl.pooled)
        g = pl.g_fc
        self.fc1.forward(pl.pooled, g, g, pl.fc1_out, relu=True)
        self.fc2.forward(pl.fc1_out, g, g, pl.fc2_out, relu=True)
        self.rcnn_pred.forward(pl.fc2_out, g, g, pl.rcnn_raw)

    # ---- backward ----------------------------------------------------------------------------------------
    def head_backward(self, pl, ws, cws):
        pyr, g = pl.pyr, pl.g_fc
        S = pl.s_rois.shape[1]
        # box head
        self._wgrad(self.rcnn_pred, pl.fc2_out, pl.d_rcnn_raw, g, g, ws, cws)
        self.rcnn_pred.dgrad(pl.d_rcnn_raw, g, g, pl.g_fc2, mask=pl.fc2_out)
        self._wgrad(self.fc2, pl.fc1_out, pl.g_fc2, g, g, ws, cws)
        self.fc2.dgrad(pl.g_fc2, g, g, pl.g_fc1, mask=pl.fc1_out)
        self._wgrad(self.fc1, pl.pooled, pl.g_fc1, g, g, ws, cws)
        self.fc1.dgrad(pl.g_fc1, g, g, pl.g_pooled)
        # first contribution to dL/dP: every pyramid level is written (zeros where no RoI sample lands, all of P6)
        if self.deterministic_roi_bwd and pl.roi_bwd_tiled:    # per-tile sums in registers, fixed order, written once (added to the RPN head's dL/dP when that ran first)
            ops.roi_align_bwd_bf16(pl.g_pooled, pyr, self.rcnn_levels, self.strides, self.fpn_ch, pl.s_rois.view(-1, 4),
                                   pl.s_labels.view(-1), S, self.pool, 2, pl.g_P, pl.roi_bwd_ws, accumulate=pl.rpn_bwd_done)
        else:
            if pl.g_feat32 is None:
                pl.g_feat32 = torch.empty((pyr.pixels, self.fpn_ch), dtype=torch.float32, device=self.device)
            if not getattr(pl, "g_feat32_clean", False):      # (normally cleared by get_losses, under the forward pass)
                pl.g_feat32.zero_()
            pl.g_feat32_clean = False
            ops.roi_align_bwd(pl.g_pooled, pyr, self.rcnn_levels, self.strides, self.fpn_ch, pl.s_rois.view(-1, 4), pl.s_labels.view(-1), S,
                              self.pool, 2, pl.g_feat32)
            ops.f32_to_bf16(pl.g_feat32, pl.g_P, accumulate=pl.rpn_bwd_done)
        # RPN head (unless get_losses already ran it under the proposal chain)
        if not pl.rpn_bwd_done:
            self._rpn_head_backward(pl, ws, cws, first=False)

    def _rpn_head_backward(self, pl, ws, cws, first):
        pyr = pl.pyr
        if self.thin_rpn_bwd and self.rpn_pred.thin_backward_ok(pyr):
            # the prediction layer (256 -> 3 + 12 channels): data, weight and bias gradient in one pass over rpn_t (csrc/conv1x1_thin.hip)
            if getattr(pl, "thin_ws", None) is None:
                pl.thin_ws = torch.empty((ops.conv1x1_thin_bwd_workspace_bytes(),), dtype=torch.uint8, device=self.device)
            self.rpn_pred.thin_backward(pl.rpn_t, pl.d_rpn_raw, pyr, pl.g_rpn_t, pl.thin_ws)
        else:
            self._wgrad(self.rpn_pred, pl.rpn_t, pl.d_rpn_raw, pyr, pyr, ws, cws)
            self.rpn_pred.dgrad(pl.d_rpn_raw, pyr, pyr, pl.g_rpn_t, mask=pl.rpn_t)
        self._wgrad(self.rpn_conv, pl.P, pl.g_rpn_t, pyr, pyr, ws, cws)
        self.rpn_conv.dgrad(pl.g_rpn_t, pyr, pyr, pl.g_P, first=first)

    def _debug_head(self, pl, out, lvl):
        A = self.num_anchors
        for i in range(pl.pyr.nlev):
            out[f"rpn_t_{i}"] = lvl(pl.rpn_t, i)
            out[f"rpn_raw_{i}"] = lvl(pl.rpn_raw, i, 5 * A)
        out["pooled"] = pl.pooled.float().cpu()
        out["fc1"] = pl.fc1_out.float().cpu()
        out["fc2"] = pl.fc2_out.float().cpu()
        out["rcnn_raw"] = pl.rcnn_raw.float().cpu()[:, : 5 * self.num_classes + 1]

    def debug_samples(self):
        """Proposals and sampled RoIs of the last forward (host copies, for the parity tests)."""
        pl = self._cur
        return dict(rois=pl.rois.cpu().numpy(), num_rois=pl.num_rois.cpu().numpy(), s_rois=pl.s_rois.cpu().numpy(),
                    s_labels=pl.s_labels.cpu().numpy(), s_targets=pl.s_targets.cpu().numpy(), s_count=pl.s_count.cpu().numpy(),
                    rpn_labels=pl.rpn_labels.cpu().numpy())

    def inference(self, inputs):
        """FasterRCNN.inference (faster_rcnn.py:98-131): RPN proposals (test top-k) -> RoIAlign + box head on every proposal ->
        softmax scores / per-class decode (rcnn.py:84-93) -> score threshold -> NMS by class -> rescale.
        The reference thresholds all R*K scores without a cap; here the NMS input is the 2048 best of them."""
        assert not self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        assert pl.N == 1, "inference supports batch size 1"
        self._cur = pl
        self.network_forward(pl)
        m = self.cfg.MODEL
        info = pre["img_info"]
        self._proposals(pl, info)
        R = pl.rois.shape[1]
        K = self.num_classes
        dev = self.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        if not hasattr(pl, "inf"):
            fin = self.fpn_ch * self.pool[0] * self.pool[1]
            pl.inf = dict(pooled=torch.empty((R, fin), **bf), fc1=torch.empty((R, 1024), **bf), fc2=torch.empty((R, 1024), **bf),
                          raw=torch.empty((R, self.rcnn_ld), **bf), g=ops.single(1, R, 1),
                          scores=torch.empty((R * K,), dtype=torch.float32, device=dev),
                          boxes=torch.empty((R * K, 4), dtype=torch.float32, device=dev))
        b = pl.inf
        ops.roi_align_fwd(pl.P, pl.pyr, self.rcnn_levels, self.strides, self.fpn_ch, pl.rois.view(-1, 4), None, R, self.pool, 2, b["pooled"])
        g = b["g"]
        self.fc1.forward(b["pooled"], g, g, b["fc1"], relu=True)
        self.fc2.forward(b["fc1"], g, g, b["fc2"], relu=True)
        self.rcnn_pred.forward(b["fc2"], g, g, b["raw"])
        ops.rcnn_predict(b["raw"], self.rcnn_ld, K, K + 1, pl.rois.view(-1, 4), pl.num_rois, R, m.RCNN_BOX_REG.MEAN, m.RCNN_BOX_REG.STD,
                         b["scores"], b["boxes"])
        return self._detect(b["scores"], [R], K, 2, info, k=2048, item_boxes=b["boxes"])
