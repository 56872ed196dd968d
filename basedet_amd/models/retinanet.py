"""RetinaNet (basedet/models/det/retinanet.py) on the HIP path: RetinaNetHead + anchor target assignment + focal / L1
losses on top of the shared ResNet-FPN trunk (fpn_base.py).

Same protocol as the reference's BaseNet (models/base_net.py:50-71): ``model(batch)`` in training mode returns
``{"total_loss", "cls_loss", "reg_loss"}``; the batch dict is the collator contract
(data/collators/pad_collator.py:38-49): ``data`` (N,3,H,W), ``gt_boxes`` (N,G,5), ``im_info`` (N,5).
"""
import math

import numpy as np
import torch
from contextlib import nullcontext as _nullcontext

from .. import ops
from ..utils.registry import registers
from . import params as P
from .fpn_base import FPNDetector, _round_up


@registers.models.register()
class RetinaNet(FPNDetector):
    @staticmethod
    def init_params(cfg, seed=0):
        return P.init_retinanet_params(cfg, seed)

    # ---- construction ------------------------------------------------------------------------------------
    def _build_head(self, add, params):
        """RetinaNetHead (layers/head/retina_head.py:9-70): two 4-conv towers + cls_score / bbox_pred, weights shared
        over the five levels."""
        m = self.cfg.MODEL
        dev = self.device
        ch = self.fpn_ch
        self.num_anchors = len(m.ANCHOR.SCALES[0]) * len(m.ANCHOR.RATIOS[0])
        nc = m.HEAD.NUM_CONVS
        self.cls_tower = [add(f"head.cls_subnet.{2 * i}", ch, ch, 3, 1, 1, bias=True) for i in range(nc)]
        self.box_tower = [add(f"head.bbox_subnet.{2 * i}", ch, ch, 3, 1, 1, bias=True) for i in range(nc)]
        A, K = self.num_anchors, self.num_classes
        self.cls_score = add("head.cls_score", ch, A * K, 3, 1, 1, bias=True)
        self.box_ld = _round_up(A * 4, 8)                                          # 36 -> 40 channels (8-aligned rows)
        self.bbox_pred = add("head.bbox_pred", ch, A * 4, 3, 1, 1, bias=True, cout_pad=self.box_ld)
        # base anchors: python float64 -> float32 (layers/common/anchor_generator.py:95-109)
        scales = np.asarray(m.ANCHOR.SCALES, np.float32).tolist()
        ratios = np.asarray(m.ANCHOR.RATIOS, np.float32).tolist()
        if len(ratios) == 1:
            ratios = ratios * len(self.strides)
        if len(scales) == 1:
            scales = scales * len(self.strides)
        self.base_anchors = []
        for sc_, ra_ in zip(scales, ratios):
            base = []
            for s_ in sc_:
                area = float(s_) ** 2.0
                for r_ in ra_:
                    w = math.sqrt(area / float(r_)); h = float(r_) * w
                    base.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
            self.base_anchors.append(torch.tensor(base, dtype=torch.float32, device=dev))

    def _head_convs(self):
        return self.cls_tower + self.box_tower + [self.cls_score, self.bbox_pred]

    def _plan_head(self, pl):
        dev = self.device
        ch = self.fpn_ch
        N = pl.N
        sizes = pl.sizes
        bf = dict(dtype=torch.bfloat16, device=dev)

        def act(g, c):
            return torch.empty((g.pixels, c), **bf)

        nc = len(self.cls_tower)
        pl.cls_act = [act(pl.pyr, ch) for _ in range(nc)]
        pl.box_act = [act(pl.pyr, ch) for _ in range(nc)]
        A, K = self.num_anchors, self.num_classes
        pl.logits = act(pl.pyr, A * K)
        pl.offsets = act(pl.pyr, self.box_ld)
        pl.d_logits = torch.empty_like(pl.logits)
        pl.d_offsets = torch.empty_like(pl.offsets)
        pl.g_tower = [[act(pl.pyr, ch) for _ in range(nc)] for _ in range(2)]   # one gradient buffer per tower layer
        # fp8 forward: every tower activation gets an e4m3 twin, written by the launch that produces it (no cast passes in the head)
        tw = lambda: torch.empty((pl.pyr.pixels, ch), dtype=torch.uint8, device=dev)      # noqa: E731
        pl.cls_act8 = [tw() if c.fp8 else None for c in self.cls_tower]
        pl.box_act8 = [tw() if c.fp8 else None for c in self.box_tower]
        # fp8 data gradients: e5m2 twins of the tower gradients and of dL/dP, written by the launch that produces them
        tg = self.fp8_grad_twins
        pl.g_tower8 = [[tw() if (c.fp8_dgrad and tg) else None for c in tower] for tower in (self.cls_tower, self.box_tower)]
        pl.g_P8 = tw() if (tg and all(t[0].fp8_dgrad for t in (self.cls_tower, self.box_tower))) else None
        # anchors (regenerated per forward in the reference, retinanet.py:116; cached per shape here)
        tot = pl.pyr.pix_per_img * A
        pl.anchors = torch.empty((tot, 4), dtype=torch.float32, device=dev)
        o = 0
        for (h, w), s, base in zip(sizes, self.strides, self.base_anchors):
            n = h * w * A
            ops.anchors_generate(h, w, s, self.cfg.MODEL.ANCHOR.OFFSET, base, pl.anchors[o:o + n])
            o += n
        pl.A_total = tot
        pl.labels = torch.empty((N, tot), dtype=torch.int32, device=dev)
        pl.match_idx = torch.empty((N, tot), dtype=torch.int32, device=dev)
        pl.gt_offsets = torch.empty((N, tot, 4), dtype=torch.float32, device=dev)
        pl.num_fg = torch.zeros((1,), dtype=torch.int32, device=dev)
        pl.loss_buf = torch.zeros((2,), dtype=torch.float32, device=dev)

    # ---- forward -----------------------------------------------------------------------------------------
    def head_forward(self, pl):
        # head (retina_head.py:103-112), all five levels per launch
        # MODEL.HEAD_TOWERS_CONCURRENT (round 6, default on): the box tower on a second stream beside the class tower -- the two are independent
        # until the losses.  Each launch is a full persistent grid, so the second tower's workgroups start on a CU the moment the first tower's
        # finish there: what is recovered is the launch-level loss of a one-workgroup-per-CU kernel (dispatch, the XCDs' finish spread:
        # the 0.907 factor of profiles/r06_pp_power.txt), +0.4-0.5 % per step on three boxes (profiles/r06_head_towers_ab.txt).  The same on the
        # BACKWARD pass, where the weight-gradient stream already runs beside the chain, costs 1.2 %: not done.  Steps that keep the weight
        # gradients on the main stream (bench.py's instrumented steps, --serial-wgrad) stay serial here too: clean per-kernel durations.
        side = None
        # (bf16 only: in fp8 mode the tower convolutions may share the main stream's cast scratch)
        # The second stream is the model's EXISTING side stream (_tstream: the target assignment at the start of the forward pass, long done
        # here; the top block's data gradients in the backward pass).  A stream of its own made five per process under torch.distributed (main,
        # weight gradients, side, communicator, this one) and the step lost 2.8 ms (577 against 643 img/s with one rank and a forced
        # all-reduce: profiles/r06_head_towers_ab.txt; cause not established -- GPU_MAX_HW_QUEUES=8 did not remove it).
        if (bool(self.cfg.MODEL.get("HEAD_TOWERS_CONCURRENT", True)) and self.device.type == "cuda" and self.async_wgrad
                and self._tstream is not None and self.weight_dtype != "fp8_e4m3"):
            side = self._tstream
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side) if side is not None else _nullcontext():
            t, t8 = pl.P, getattr(pl, "P8", None)
            for c, a, a8 in zip(self.box_tower, pl.box_act, pl.box_act8):
                c.forward(t, pl.pyr, pl.pyr, a, relu=True, x8=t8, y8=a8); t, t8 = a, a8
            self.bbox_pred.forward(t, pl.pyr, pl.pyr, pl.offsets, x8=t8)
        t, t8 = pl.P, getattr(pl, "P8", None)
        for c, a, a8 in zip(self.cls_tower, pl.cls_act, pl.cls_act8):
            c.forward(t, pl.pyr, pl.pyr, a, relu=True, x8=t8, y8=a8); t, t8 = a, a8
        self.cls_score.forward(t, pl.pyr, pl.pyr, pl.logits, x8=t8)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    def get_losses(self, inputs):
        """RetinaNet.get_losses (retinanet.py:120-170)."""
        assert self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        self._cur = pl
        m = self.cfg.MODEL
        gt = pre["gt_boxes"]
        num_gt = pre["img_info"][:, 4].to(torch.int32).contiguous()
        N, Gmax = gt.shape[0], gt.shape[1]
        ws = pl.wgrad_ws[: N * Gmax]
        thr = m.MATCHER.THRESHOLDS
        # The target assignment depends on the anchors and the gt boxes only (retinanet.py:211-232), not on the network's output: its two
        # launches (~0.12 ms at 16 x 201 600 anchors) run on a side stream under the forward pass instead of between forward and losses.
        side = self._tstream if (self.async_wgrad and self._tstream is not None and m.get("ASSIGN_ON_SIDE_STREAM", True)) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side) if side is not None else _nullcontext():
            ops.retina_assign_encode(pl.anchors, gt, num_gt, thr[0], thr[1], m.MATCHER.ALLOW_LOW_QUALITY, m.BOX_REG.MEAN,
                                     m.BOX_REG.STD, pl.labels, pl.match_idx, pl.gt_offsets, pl.num_fg, ws)
        self.network_forward(pl)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        pl.loss_buf.zero_()
        rows = N * pl.A_total
        ops.focal_loss_fwd_bwd(pl.logits, pl.labels, rows, self.num_classes, m.LOSSES.FOCAL_LOSS_ALPHA,
                               m.LOSSES.FOCAL_LOSS_GAMMA, pl.num_fg, 1.0, pl.loss_buf[0:1], pl.d_logits)
        ops.smooth_l1_fwd_bwd(pl.offsets, pl.gt_offsets, pl.labels, pl.pyr.pixels, self.num_anchors, self.box_ld,
                              m.LOSSES.SMOOTH_L1_BETA, pl.num_fg, m.LOSSES.REG_LOSS_WEIGHT, pl.loss_buf[1:2], pl.d_offsets)
        cls_loss, reg_loss = pl.loss_buf[0], pl.loss_buf[1]
        return {"total_loss": cls_loss + reg_loss, "cls_loss": cls_loss, "reg_loss": reg_loss}

    # ---- backward ----------------------------------------------------------------------------------------
    def head_backward(self, pl, ws, cws):
        pyr = pl.pyr
        # ---- head: cls tower then box tower; both end in g_P.  g_tower[t][i] = dL/d(pre-activation of tower conv i)
        # fp8 mode: a tower gradient's e5m2 twin g8[i] is written by the data-gradient launch that produces it and read by the next
        # one (no cast passes inside a tower); a twin is valid only if its producer writes one (the 40-channel box regressor does not)
        for ti, (tower, acts, pred, dpred) in enumerate(((self.cls_tower, pl.cls_act, self.cls_score, pl.d_logits),
                                                         (self.box_tower, pl.box_act, self.bbox_pred, pl.d_offsets))):
            gbuf, g8 = pl.g_tower[ti], pl.g_tower8[ti]
            n = len(tower)
            gs = tower[n - 1].grad_scale
            self._wgrad(pred, acts[-1], dpred, pyr, pyr, ws, cws)
            tw = g8[n - 1] if pred.dgrad_writes_twin(pyr, pyr) else None       # None also when twins are off (g8 holds no buffers)
            if not pred.dgrad(dpred, pyr, pyr, gbuf[n - 1], mask=acts[-1], dx8=tw, q_scale=gs):
                tw = None
            act8 = pl.cls_act8 if ti == 0 else pl.box_act8
            for i in range(n - 1, -1, -1):
                x = acts[i - 1] if i > 0 else pl.P
                # the weight gradient reads the same twins: the tower input's (forward) and the gradient's (tw: written with gbuf[i])
                x8 = (act8[i - 1] if i > 0 else pl.P8) if tw is not None else None
                self._wgrad(tower[i], x, gbuf[i], pyr, pyr, ws, cws, x8=x8, g8=tw)
                if i > 0:
                    nxt = g8[i - 1] if tower[i].dgrad_writes_twin(pyr, pyr) else None
                    wrote = tower[i].dgrad(gbuf[i], pyr, pyr, gbuf[i - 1], mask=acts[i - 1], g8=tw, dx8=nxt, q_scale=tower[i - 1].grad_scale)
                    tw = nxt if wrote else None
                else:
                    # dL/dP's twin goes to the FPN output convolutions: THEIR scale (another scale group)
                    wrote_p = tower[i].dgrad(gbuf[i], pyr, pyr, pl.g_P, first=(ti == 0), g8=tw,
                                             dx8=pl.g_P8 if tower[i].dgrad_writes_twin(pyr, pyr) else None,
                                             q_scale=self.output[self.fpn_stages[0]].grad_scale)
        # the box tower's launch wrote the twin of the FINAL dL/dP (it accumulates onto the class tower's contribution)
        pl.g_P8_ready = pl.g_P8 is not None and bool(wrote_p)

    def _debug_head(self, pl, out, lvl):
        for i in range(pl.pyr.nlev):
            for k in range(len(self.cls_tower)):
                out[f"cls{k}_{i}"] = lvl(pl.cls_act[k], i)
                out[f"box{k}_{i}"] = lvl(pl.box_act[k], i)
            out[f"logits_{i}"] = lvl(pl.logits, i)
            out[f"offs_{i}"] = lvl(pl.offsets, i, self.num_anchors * 4)

    # ------------------------------------------------------------------------------------------------
    # inference (retinanet.py:172-201) -- single image, like the reference
    # ------------------------------------------------------------------------------------------------
    def inference(self, inputs):
        assert not self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        assert pl.N == 1, "inference supports batch size 1 (retinanet.py:174)"
        self.network_forward(pl)
        K, A = self.num_classes, self.num_anchors
        m = self.cfg.MODEL
        rows = pl.pyr.pix_per_img * A                               # anchors
        scores = torch.empty((rows * K,), dtype=torch.float32, device=self.device)
        ops.det_scores(pl.logits, rows, K, scores)                  # F.sigmoid(F.flatten(logits)) (:184)
        return self._detect(scores, [h * w * A for h, w in pl.sizes], K, 0, pre["img_info"], anchors=pl.anchors, offsets=pl.offsets,
                            off_ld=self.box_ld, A=A, mean=m.BOX_REG.MEAN, std=m.BOX_REG.STD)
