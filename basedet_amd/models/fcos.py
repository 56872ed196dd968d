"""FCOS (basedet/models/det/fcos.py) on the HIP path: PointHead (GroupNorm towers, per-level scales, centre-ness) +
point target assignment + focal / GIoU / BCE losses on the shared ResNet-FPN trunk.

``model(batch)`` returns ``{"total_loss", "cls_loss", "reg_loss", "ctr_loss"}`` (fcos.py:173-178).
Data-parallel detail (fcos.py:143-144): ``num_fg`` and ``sum_ctr`` are all-reduced (mean) across ranks before the losses
are normalised -- one 2-float RCCL all-reduce in the forward pass.
"""
import torch
from contextlib import nullcontext as _nullcontext

from .. import comm as _comm
from .. import ops
from ..utils.registry import registers
from . import params as P
from .engine import FusedPredConv
from .fpn_base import FPNDetector, VecParam

GN_EPS = 1e-5   # megengine.module.normalization.GroupNorm default


@registers.models.register()
class FCOS(FPNDetector):
    ASSIGN_ON_SIDE_STREAM = False      # target assignment on a side stream under the forward pass (get_losses); ATSS: True

    @staticmethod
    def init_params(cfg, seed=0):
        return P.init_fcos_params(cfg, seed)

    # ---- construction ------------------------------------------------------------------------------------
    def _build_head(self, add, params):
        """PointHead (layers/head/point_head.py:40-105)."""
        m = self.cfg.MODEL
        ch = self.fpn_ch
        assert m.ANCHOR.NUM_ANCHORS == 1, "the HIP PointHead path supports one anchor point per location"
        assert ch == 256, "GroupNorm kernel: 32 groups x 8 channels"
        nc = m.HEAD.NUM_CONVS
        self.towers = {}
        for tower in ("cls_subnet", "bbox_subnet"):
            convs, gammas, betas = [], [], []
            for i in range(nc):
                convs.append(add(f"head.{tower}.{3 * i}", ch, ch, 3, 1, 1, bias=True))
                g = VecParam(f"head.{tower}.{3 * i + 1}.weight", ch)
                b = VecParam(f"head.{tower}.{3 * i + 1}.bias", ch)
                self.vparams[g.name] = g
                self.vparams[b.name] = b
                gammas.append(g); betas.append(b)
            self.towers[tower] = (convs, gammas, betas)
        self.cls_score = add("head.cls_score", ch, self.num_classes, 3, 1, 1, bias=True)
        # bbox_pred (4) and ctrness (1) read the same tower output: one conv with 5 (padded to 8) output channels
        self.pred = FusedPredConv("head.bbox_ctr", [("head.bbox_pred", 4), ("head.ctrness", 1)], ch, 3, 1, 1, self.device, cout_pad=8)
        self.convs[self.pred.name] = self.pred
        self.scales = VecParam("head.scales", len(self.strides))
        self.vparams[self.scales.name] = self.scales

    def _head_convs(self):
        return self.towers["cls_subnet"][0] + self.towers["bbox_subnet"][0] + [self.cls_score, self.pred]

    def _plan_head(self, pl):
        dev = self.device
        ch = self.fpn_ch
        N = pl.N
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)

        def act(c):
            return torch.empty((pl.pyr.pixels, c), **bf)

        nc = len(self.towers["cls_subnet"][0])
        L = pl.pyr.nlev
        pl.tw = {}
        for tower in self.towers:
            pl.tw[tower] = dict(y=[act(ch) for _ in range(nc)], z=[act(ch) for _ in range(nc)],
                                stats=[torch.empty((N, L, 32, 2), **f32) for _ in range(nc)])
        pl.logits = act(self.num_classes)
        pl.raw = act(8)
        pl.offsets = act(4)
        pl.d_logits = torch.empty_like(pl.logits)
        pl.d_off = torch.empty_like(pl.offsets)
        pl.d_ctr = torch.empty((pl.pyr.pixels,), **bf)
        pl.d_raw = torch.empty_like(pl.raw)
        pl.g_z = act(ch)                                                  # dL/dz scratch (consumed at once by the GN bwd)
        pl.g_y = [[act(ch) for _ in range(nc)] for _ in range(2)]         # dL/dy per tower layer (read by async wgrads)
        P_total = pl.pyr.pix_per_img
        pl.points = torch.empty((P_total, 2), **f32)
        o = 0
        for (h, w), s in zip(pl.sizes, self.strides):
            ops.points_generate(h, w, s, self.cfg.MODEL.ANCHOR.OFFSET, 1, pl.points[o:o + h * w])
            o += h * w
        pl.lvl_start = [0]
        for (h, w) in pl.sizes:
            pl.lvl_start.append(pl.lvl_start[-1] + h * w)
        pl.labels = torch.empty((N, P_total), dtype=torch.int32, device=dev)
        pl.gt_offsets = torch.empty((N, P_total, 4), **f32)
        pl.gt_ctr = torch.empty((N, P_total), **f32)
        pl.stats = torch.zeros((2,), **f32)          # (num_fg, sum_ctr)
        pl.loss_buf = torch.zeros((3,), **f32)
        pl.gn_ws = torch.empty((ops.groupnorm_workspace_bytes(N, L, ch, pl.pyr.pix_per_img) // 4 + 16,), **f32)
        pl.off_ws = torch.empty((ops.fcos_offsets_workspace_bytes() // 4,), **f32)
        # MODEL.FUSE_GN_STATS (default on): GroupNorm's statistics from the tower convolutions' epilogues instead of a pass over their outputs
        tower_convs = self.towers["cls_subnet"][0] + self.towers["bbox_subnet"][0]
        fuse = bool(self.cfg.MODEL.get("FUSE_GN_STATS", True)) and dev.type == "cuda" and all(c.gnstats_ok() for c in tower_convs)
        pl.gn_part = (torch.empty((ops.conv2d_fwd_gnstats_bytes(tower_convs[0].desc(pl.pyr, pl.pyr)) // 4,), **f32) if fuse else None)

    # ---- forward -----------------------------------------------------------------------------------------
    def head_forward(self, pl):
        """PointHead.forward (point_head.py:137-151), all levels per launch."""
        pyr, ch = pl.pyr, self.fpn_ch
        for tower, (convs, gammas, betas) in self.towers.items():
            t = pl.P
            bufs = pl.tw[tower]
            for i, c in enumerate(convs):
                if pl.gn_part is not None:
                    # the statistics pass rides in the convolution's epilogue (round 6: bd_conv2d_fwd_gnstats / bd_groupnorm_fwd_parts)
                    c.forward_gnstats(t, pyr, pyr, bufs["y"][i], pl.gn_part)
                    ops.groupnorm_fwd_parts(c.desc(pyr, pyr), bufs["y"][i], pl.gn_part, gammas[i].w, betas[i].w, GN_EPS, True, bufs["stats"][i],
                                            bufs["z"][i])
                else:
                    c.forward(t, pyr, pyr, bufs["y"][i])
                    ops.groupnorm_fwd(bufs["y"][i], gammas[i].w, betas[i].w, pyr, ch, GN_EPS, True, bufs["stats"][i], bufs["z"][i], pl.gn_ws)
                t = bufs["z"][i]
            if tower == "cls_subnet":
                self.cls_score.forward(t, pyr, pyr, pl.logits)
            else:
                self.pred.forward(t, pyr, pyr, pl.raw)
        ops.fcos_offsets_fwd(pl.raw, 8, self.scales.w, pyr, self.strides, pl.offsets)     # relu(x * scale_l) * stride_l (:143)

    def get_losses(self, inputs):
        """FCOS.get_losses (fcos.py:114-179)."""
        assert self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        self._cur = pl
        m = self.cfg.MODEL
        gt = pre["gt_boxes"]
        num_gt = pre["img_info"][:, 4].to(torch.int32).contiguous()
        # FCOS / ATSS targets depend on the points and the gt boxes only (fcos.py:222-293, atss.py:17-86): they can be assigned on a side
        # stream under the forward pass (as RetinaNet's).  Measured, same box: ATSS (two launches per (gt, image)) +0.5 %, FCOS (one cheap
        # launch: the two stream joins cost more than it) -0.3 % -- on for ATSS, off for FCOS.
        side = self._tstream if (self.async_wgrad and self._tstream is not None and m.get("ASSIGN_ON_SIDE_STREAM", self.ASSIGN_ON_SIDE_STREAM)) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side) if side is not None else _nullcontext():
            self._assign(pl, gt, num_gt)
        self.network_forward(pl)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        c = _comm.get_comm()
        if c is not None:                                              # a one-rank communicator (BD_FORCE_ALLREDUCE) goes through RCCL too: identity
            self._allreduce_stats(c, pl.stats)                         # all_reduce(mode="mean") of num_fg and sum_ctr (fcos.py:143-144)
        pl.loss_buf.zero_()
        rows = pl.N * pl.pyr.pix_per_img
        assert m.LOSSES.IOU_LOSS_TYPE == "giou", "HIP FCOS path implements the giou ltrb loss"
        ops.focal_loss_fwd_bwd(pl.logits, pl.labels, rows, self.num_classes, m.LOSSES.FOCAL_LOSS_ALPHA, m.LOSSES.FOCAL_LOSS_GAMMA,
                               pl.stats[0:1], 1.0, pl.loss_buf[0:1], pl.d_logits)
        ops.giou_ltrb_fwd_bwd(pl.offsets, pl.gt_offsets, pl.gt_ctr, pl.labels, rows, pl.stats[1:2], m.LOSSES.REG_LOSS_WEIGHT,
                              pl.loss_buf[1:2], pl.d_off)
        ops.bce_logits_fwd_bwd(pl.raw, pl.gt_ctr, pl.labels, rows, pl.stats[0:1], pl.loss_buf[2:3], pl.d_ctr, ld=8, off=4)
        cls_loss, reg_loss, ctr_loss = pl.loss_buf[0], pl.loss_buf[1], pl.loss_buf[2]
        return {"total_loss": cls_loss + reg_loss + ctr_loss, "cls_loss": cls_loss, "reg_loss": reg_loss, "ctr_loss": ctr_loss}

    @staticmethod
    def _allreduce_stats(c, stats):
        """The two-scalar mean over the ranks on the communicator's OWN stream (after this stream's work so far; this stream then
        waits for it): every collective of a communicator is issued on one stream, in the same order on every rank -- the bucket
        all-reduces of the previous backward, this exchange, the buckets of this step's backward."""
        prod = [torch.cuda.current_stream()] if stats.is_cuda else []
        c.allreduce_async(stats, prod, "avg")
        c.wait()

    def _assign(self, pl, gt, num_gt):
        """FCOS.get_ground_truth (fcos.py:222-293)."""
        m = self.cfg.MODEL
        ops.fcos_assign(pl.points, pl.lvl_start, m.HEAD.OBJECT_SIZES_OF_INTEREST, self.strides, m.HEAD.CENTER_SAMPLING_RADIUS,
                        gt, num_gt, pl.labels, pl.gt_offsets, pl.gt_ctr, pl.stats)

    # ---- backward ----------------------------------------------------------------------------------------
    def head_backward(self, pl, ws, cws):
        pyr, ch = pl.pyr, self.fpn_ch
        ops.fcos_offsets_bwd(pl.raw, 8, self.scales.w, pyr, self.strides, pl.d_off, pl.d_ctr, pl.d_raw, self.scales.g, pl.off_ws)
        for ti, (tower, pred, dpred) in enumerate((("cls_subnet", self.cls_score, pl.d_logits), ("bbox_subnet", self.pred, pl.d_raw))):
            convs, gammas, betas = self.towers[tower]
            bufs = pl.tw[tower]
            nc = len(convs)
            self._wgrad(pred, bufs["z"][nc - 1], dpred, pyr, pyr, ws, cws)
            pred.dgrad(dpred, pyr, pyr, pl.g_z)             # gradient w.r.t. z (the ReLU gate is applied in the GroupNorm bwd)
            for i in range(nc - 1, -1, -1):
                gy = pl.g_y[ti][i]
                # g_z = dL/dz_i  ->  gy = dL/dy_i (conv output)
                ops.groupnorm_bwd(pl.g_z, bufs["y"][i], gammas[i].w, betas[i].w, bufs["stats"][i], pyr, ch, True, gy,
                                  gammas[i].g, betas[i].g, pl.gn_ws)
                x = bufs["z"][i - 1] if i > 0 else pl.P
                self._wgrad(convs[i], x, gy, pyr, pyr, ws, cws)
                if i > 0:
                    convs[i].dgrad(gy, pyr, pyr, pl.g_z)
                else:
                    convs[i].dgrad(gy, pyr, pyr, pl.g_P, first=(ti == 0))

    def _debug_head(self, pl, out, lvl):
        for i in range(pl.pyr.nlev):
            for tower, tag in (("cls_subnet", "cls"), ("bbox_subnet", "box")):
                for k in range(len(self.towers[tower][0])):
                    out[f"{tag}y{k}_{i}"] = lvl(pl.tw[tower]["y"][k], i)
                    out[f"{tag}{k}_{i}"] = lvl(pl.tw[tower]["z"][k], i)
            out[f"logits_{i}"] = lvl(pl.logits, i)
            out[f"raw_{i}"] = lvl(pl.raw, i, 5)

    def inference(self, inputs):
        """FCOS.inference (fcos.py:181-216): score = sqrt(sigmoid(cls) * sigmoid(ctrness)), PointCoder.decode."""
        assert not self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        assert pl.N == 1, "inference supports batch size 1 (fcos.py:183)"
        self.network_forward(pl)
        K = self.num_classes
        rows = pl.pyr.pix_per_img
        scores = torch.empty((rows * K,), dtype=torch.float32, device=self.device)
        ops.det_scores(pl.logits, rows, K, scores, ctr=pl.raw, ctr_ld=8, ctr_off=4)
        return self._detect(scores, [h * w for h, w in pl.sizes], K, 1, pre["img_info"], anchors=pl.points, offsets=pl.offsets,
                            off_ld=4, A=1)


@registers.models.register()
class ATSS(FCOS):
    """ATSS (basedet/models/det/atss.py): the FCOS network and losses with the adaptive training-sample selection."""

    ASSIGN_ON_SIDE_STREAM = True

    def _assign(self, pl, gt, num_gt):
        m = self.cfg.MODEL
        N, P_total = pl.labels.shape
        if getattr(pl, "atss_ws", None) is None:
            pl.atss_ws = torch.empty((ops.atss_assign_workspace_bytes(N, P_total),), dtype=torch.uint8, device=self.device)
        ops.atss_assign(pl.points, pl.lvl_start, self.strides, m.ANCHOR.TOPK, m.ANCHOR.SCALE, gt, num_gt, pl.labels, pl.gt_offsets,
                        pl.gt_ctr, pl.stats, pl.atss_ws)


@registers.models.register()
class OTA(FCOS):
    """OTA (basedet/models/det/ota.py): the FCOS network -- OTAPointHead with NORM_REG_TARGETS is PointHead's forward, its
    centre-ness branch read as the IoU prediction (point_head.py:154-212) -- with the prediction-aware dynamic top-k assignment
    (bd_ota_assign) and emd_losses' weighting (:183-233): focal / num_fg, 2 x GIoU / num_fg, 0.5 x BCE(iou) / num_fg."""

    def __init__(self, cfg, *a, **k):
        self.matching = cfg.MODEL.get("MATCHING", "topk")
        assert self.matching in ("topk", "sinkhorn"), f"unsupported matching named {self.matching}"      # ota.py:21
        hc = cfg.MODEL.HEAD
        assert hc.get("NORM_REG_TARGETS", True) and hc.get("WITH_NORM", True) and hc.get("SHARE_PARAM", True), \
            "OTA on the HIP path supports the reference's OTAConfig head flags (NORM_REG_TARGETS, WITH_NORM, SHARE_PARAM)"
        super().__init__(cfg, *a, **k)

    def _plan_head(self, pl):
        super()._plan_head(pl)
        pl.ota_ws = torch.empty((ops.ota_assign_workspace_bytes(pl.N, pl.points.shape[0]),), dtype=torch.uint8, device=self.device)

    def _assign(self, pl, gt, num_gt):
        m = self.cfg.MODEL
        if self.matching == "sinkhorn":              # SinkhornMatcher(eps=0.1, max_iter=50) (ota.py:43-44)
            need = ops.ota_sinkhorn_workspace_bytes(pl.N, pl.points.shape[0], gt.shape[1])
            if pl.ota_ws.numel() < need:
                pl.ota_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
            ops.ota_assign_sinkhorn(pl.points, pl.lvl_start, self.strides, pl.logits, self.num_classes, pl.offsets, gt, num_gt,
                                    m.LOSSES.FOCAL_LOSS_ALPHA, m.LOSSES.FOCAL_LOSS_GAMMA, m.HEAD.get("COST_REG_WEIGHTS", 1.5), 2.5,
                                    pl.labels, pl.gt_offsets, pl.gt_ctr, pl.stats, pl.ota_ws)
            return
        ops.ota_assign(pl.points, pl.lvl_start, self.strides, pl.logits, self.num_classes, pl.offsets, gt, num_gt,
                       m.LOSSES.FOCAL_LOSS_ALPHA, m.LOSSES.FOCAL_LOSS_GAMMA, m.HEAD.get("COST_REG_WEIGHTS", 1.5), 2.5,
                       m.HEAD.get("CANDIDATE_K", 10), pl.labels, pl.gt_offsets, pl.gt_ctr, pl.stats, pl.ota_ws)

    def get_losses(self, inputs):
        """OTA.get_losses (ota.py:62-74) + emd_losses (:183-233); pl.gt_ctr holds the IoU targets, pl.stats = (num_fg, 2 num_fg)."""
        assert self.training
        pre = self.pre_process(inputs)
        pl = pre["plan"]
        self._cur = pl
        self.network_forward(pl)
        m = self.cfg.MODEL
        gt = pre["gt_boxes"]
        num_gt = pre["img_info"][:, 4].to(torch.int32).contiguous()
        self._assign(pl, gt, num_gt)
        c = _comm.get_comm()
        if c is not None:                                              # a one-rank communicator (BD_FORCE_ALLREDUCE) goes through RCCL too: identity
            self._allreduce_stats(c, pl.stats)                         # all_reduce(num_foreground, mode="mean") (ota.py:200)
        pl.loss_buf.zero_()
        rows = pl.N * pl.pyr.pix_per_img
        assert m.LOSSES.IOU_LOSS_TYPE == "giou", "HIP OTA path implements the giou ltrb loss"
        ops.focal_loss_fwd_bwd(pl.logits, pl.labels, rows, self.num_classes, m.LOSSES.FOCAL_LOSS_ALPHA, m.LOSSES.FOCAL_LOSS_GAMMA,
                               pl.stats[0:1], 1.0, pl.loss_buf[0:1], pl.d_logits)
        ops.giou_ltrb_fwd_bwd(pl.offsets, pl.gt_offsets, None, pl.labels, rows, pl.stats[0:1], 2.0, pl.loss_buf[1:2], pl.d_off)
        ops.bce_logits_fwd_bwd(pl.raw, pl.gt_ctr, pl.labels, rows, pl.stats[1:2], pl.loss_buf[2:3], pl.d_ctr, ld=8, off=4)
        loss_cls, loss_box, loss_iou = pl.loss_buf[0], pl.loss_buf[1], pl.loss_buf[2]
        return {"total_loss": loss_cls + loss_box + loss_iou, "loss_cls": loss_cls, "loss_offsets": loss_box, "loss_ious": loss_iou}
