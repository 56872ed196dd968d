"""CPU ORACLE (test infrastructure only) -- torch-CPU fp32 restatement of the reference's
RetinaNet / FCOS training step (backbone + FPN + head + target assignment + losses + SGD).

NOT part of the product path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this.  PARITY UNPINNED: the reference's model tests assert no values
(tests/models/test_retinanet.py:18-31) and megengine/basecore cannot be imported here, so this
restatement (with the documented choices below) is the only oracle for conv/loss values.

Documented choices where the reference delegates to un-vendored code:
  * FrozenBN (basecore ``get_norm("FrozenBN")``, layers/backbone/build.py:23): y = x*scale + shift with
    scale = weight / sqrt(running_var + 1e-5), shift = bias - running_mean*scale; no gradients.
  * ``F.nn.interpolate(scale_factor=2, mode="BILINEAR")`` (fpn_backbone.py:143): align_corners=False,
    source index clamped at the border (PyTorch semantics).
  * SGD (megengine.optimizer.SGD): g' = g + wd*w; v = momentum*v + g'; w -= lr*v.
Parameters are passed in as a dict name -> numpy array, names follow the reference's state_dict
(``backbone.bottom_up.layer2.0.conv1.weight`` ...), conv weights OIHW, 1-D bias/BN vectors.
"""
import numpy as np
import torch
import torch.nn.functional as TF

from . import box_ops

BN_EPS = 1e-5

RESNET_SPECS = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


class _Inject(torch.autograd.Function):
    """Forward: return the injected activation (taken from the HIP run); backward: the ReLU gate of that
    activation.  Turns the oracle backward into the exact linear map the HIP backward must implement."""

    @staticmethod
    def forward(ctx, x, inj, relu):
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(inj > 0)
        return inj.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.relu:
            (m,) = ctx.saved_tensors
            g = g * m
        return g, None, None


class Oracle:
    """Functional model over a parameter dict.  ``trainable`` names get requires_grad."""

    def __init__(self, params, arch, trainable=(), sim_bf16=False, inject=None, record=None):
        """sim_bf16: round weights (after folding the FrozenBN scale) and every stored activation to bf16 with a
        straight-through gradient -- the rounding points of the HIP path -- while all arithmetic stays fp32.
        With it the ReLU masks of both sides coincide, which makes the gradient comparison tight."""
        self.sim_bf16 = sim_bf16
        self.inject = inject      # dict key -> NCHW tensor: stored activations of the HIP run (see _act)
        self.record = record      # optional dict: key -> (norm, squared error against `inject`-style reference, count); see _act
        self.arch = dict(arch)
        self.p = {}
        tset = set(trainable)
        for k, v in params.items():
            t = _t(v).clone()
            if k in tset:
                t.requires_grad_(True)
            self.p[k] = t
        self.trainable = [k for k in params if k in tset]

    # ---- primitives -------------------------------------------------------------------------
    def _act(self, key, x, relu):
        """Every tensor the HIP path stores goes through here: (ReLU), bf16 rounding, optional injection."""
        if self.inject is not None and key in self.inject:
            return _Inject.apply(x, self.inject[key], relu)
        y = self._q(TF.relu(x) if relu else x)
        if self.record is not None and isinstance(self.record.get("_compare"), dict) and key in self.record["_compare"]:
            # per-layer forward comparison (tests): rel-L2 of the other side's stored activation against this fp32 value
            ref = y.detach().double()
            got = self.record["_compare"][key].double()
            self.record[key] = float((got - ref).norm() / (ref.norm() + 1e-30))
        return y

    def _q(self, x):
        if not self.sim_bf16:
            return x
        return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()

    def _bn_scale_shift(self, prefix):
        p = self.p
        scale = p[prefix + ".weight"] / torch.sqrt(p[prefix + ".running_var"] + BN_EPS)
        shift = p[prefix + ".bias"] - p[prefix + ".running_mean"] * scale
        return scale, shift

    def _bn(self, x, prefix):
        scale, shift = self._bn_scale_shift(prefix)
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)

    def _conv(self, x, name, stride=1, pad=0):
        b = self.p.get(name + ".bias")
        return TF.conv2d(x, self._q(self.p[name + ".weight"]), b, stride=stride, padding=pad)

    def _conv_bn(self, x, name, bn, stride=1, pad=0):
        """conv + FrozenBN; in sim_bf16 mode the scale is folded into the weight before rounding (HIP path)."""
        if not self.sim_bf16:
            return self._bn(self._conv(x, name, stride, pad), bn)
        scale, shift = self._bn_scale_shift(bn)
        w = self._q(self.p[name + ".weight"] * scale.view(-1, 1, 1, 1))
        return TF.conv2d(x, w, shift, stride=stride, padding=pad)

    # ---- backbone (models/cls/resnet.py) -----------------------------------------------------
    def _bottleneck(self, x, pre, stride, has_ds):
        idt = x
        y = self._act(pre + ".a0", self._conv_bn(x, pre + ".conv1", pre + ".bn1"), True)
        y = self._act(pre + ".a1", self._conv_bn(y, pre + ".conv2", pre + ".bn2", stride, 1), True)   # stride on the 3x3 (:72-81)
        y = self._conv_bn(y, pre + ".conv3", pre + ".bn3")
        if has_ds:
            idt = self._act(pre + ".idt", self._conv_bn(x, pre + ".downsample.0", pre + ".downsample.1", stride), False)
        return self._act(pre + ".out", y + idt, True)

    def _basic(self, x, pre, stride, has_ds):
        idt = x
        y = self._act(pre + ".a0", self._conv_bn(x, pre + ".conv1", pre + ".bn1", stride, 1), True)
        y = self._conv_bn(y, pre + ".conv2", pre + ".bn2", 1, 1)
        if has_ds:
            idt = self._act(pre + ".idt", self._conv_bn(x, pre + ".downsample.0", pre + ".downsample.1", stride), False)
        return self._act(pre + ".out", y + idt, True)

    def backbone(self, x):
        """ResNet.extract_features (resnet.py:236-252)."""
        bu = "backbone.bottom_up"
        kind, layers = RESNET_SPECS[self.arch["backbone"]]
        x = self._q(TF.relu(self._conv_bn(self._q(x), bu + ".conv1", bu + ".bn1", 2, 3)))
        x = TF.max_pool2d(x, 3, 2, 1)
        if self.inject is not None and "pool" in self.inject:
            x = self.inject["pool"].clone()
        outs = {"stem": x}
        block = self._bottleneck if kind == "bottleneck" else self._basic
        for li, nblk in enumerate(layers):
            for b in range(nblk):
                stride = 2 if (b == 0 and li > 0) else 1
                has_ds = (bu + f".layer{li + 1}.{b}.downsample.0.weight") in self.p
                x = block(x, bu + f".layer{li + 1}.{b}", stride, has_ds)
            outs[f"res{li + 2}"] = x
        return outs

    def fpn(self, feats):
        """FPN.forward + LastLevelP6P7 (fpn_backbone.py:123-160, 198-204)."""
        names = self.arch.get("fpn_in", ["res3", "res4", "res5"])
        stages = [int(n[-1]) for n in names]
        x = [feats[n] for n in names[::-1]]
        st = stages[::-1]
        q, act = self._q, self._act
        prev = act(f"lat{st[0]}", self._conv(x[0], f"backbone.fpn_lateral{st[0]}"), False)
        results = [act(f"P{st[0]}", self._conv(prev, f"backbone.fpn_output{st[0]}", 1, 1), False)]
        for f, s in zip(x[1:], st[1:]):
            td = TF.interpolate(prev, scale_factor=2, mode="bilinear", align_corners=False)
            prev = act(f"lat{s}", q(self._conv(f, f"backbone.fpn_lateral{s}")) + td, False)
            results.insert(0, act(f"P{s}", self._conv(prev, f"backbone.fpn_output{s}", 1, 1), False))
        top = stages[-1]
        if self.arch.get("top_block") == "pool":          # FPNP6 (fpn_backbone.py:172-183)
            return results + [TF.max_pool2d(results[-1], kernel_size=1, stride=2, padding=0)]
        p6 = act(f"P{top + 1}", self._conv(feats["res5"], "backbone.top_block.p6", 2, 1), False)
        p7 = act(f"P{top + 2}", self._conv(TF.relu(p6), "backbone.top_block.p7", 2, 1), False)
        return results + [p6, p7]

    # ---- heads --------------------------------------------------------------------------------
    def retina_head(self, feats):
        """RetinaNetHead.forward (retina_head.py:103-112); Sequential indices 0,2,4,6 are the convs."""
        logits, offsets = [], []
        nconv = self.arch.get("num_convs", 4)
        for li, f in enumerate(feats):
            c = f
            b = f
            for i in range(nconv):
                c = self._act(f"cls{i}_{li}", self._conv(c, f"head.cls_subnet.{2 * i}", 1, 1), True)
                b = self._act(f"box{i}_{li}", self._conv(b, f"head.bbox_subnet.{2 * i}", 1, 1), True)
            logits.append(self._act(f"logits_{li}", self._conv(c, "head.cls_score", 1, 1), False))
            offsets.append(self._act(f"offs_{li}", self._conv(b, "head.bbox_pred", 1, 1), False))
        return logits, offsets

    @staticmethod
    def _permute(t, K):
        """permute_to_N_Any_K (layers/common/function.py:26-32)."""
        n = t.shape[0]
        return t.permute(0, 2, 3, 1).reshape(n, -1, K)

    # ---- RetinaNet.get_losses (models/det/retinanet.py:120-170) --------------------------------
    def retinanet_forward(self, image):
        feats = self.fpn(self.backbone(image))
        logits, offsets = self.retina_head(feats)
        K = self.arch["num_classes"]
        logits = torch.cat([self._permute(x, K) for x in logits], dim=1)
        offsets = torch.cat([self._permute(x, 4) for x in offsets], dim=1)
        sizes = [tuple(f.shape[-2:]) for f in feats]
        return logits, offsets, sizes, feats

    def retinanet_losses(self, batch):
        a = self.arch
        image = _t(box_ops.data_to_input(batch["data"], a["img_mean"], a["img_std"]))
        logits, offsets, sizes, _ = self.retinanet_forward(image)
        anchors = np.concatenate(box_ops.default_anchors(sizes, a["strides"], a["anchor_scales"], a["anchor_ratios"], a["anchor_offset"]), 0)
        num_valid = np.asarray(batch["im_info"])[:, 4].astype(np.int32)
        labels, gt_off, _ = box_ops.retinanet_ground_truth(anchors, batch["gt_boxes"], num_valid)
        K = a["num_classes"]
        logits = logits.reshape(-1, K)
        offsets = offsets.reshape(-1, 4)
        labels_t = torch.from_numpy(labels.reshape(-1)).long()
        gt_off_t = _t(gt_off.reshape(-1, 4))
        valid = labels_t >= 0
        fg = labels_t > 0
        num_fg = int(fg.sum())
        tgt = torch.zeros_like(logits)
        tgt[fg, labels_t[fg] - 1] = 1
        x = logits[valid]
        t = tgt[valid]
        p = torch.sigmoid(x)
        ce = -(t * TF.logsigmoid(x) + (1 - t) * TF.logsigmoid(-x))
        fl = ce * (t * (1 - p) + (1 - t) * p) ** a.get("focal_gamma", 2.0)
        al = a.get("focal_alpha", 0.25)
        fl = fl * (t * al + (1 - t) * (1 - al))
        cls_loss = fl.sum() / max(1, num_fg)
        d = offsets[fg] - gt_off_t[fg]
        beta = a.get("smooth_l1_beta", 0.0)
        if beta < 1e-5:
            l1 = d.abs()
        else:
            l1 = torch.where(d.abs() < beta, 0.5 * d ** 2 / beta, d.abs() - 0.5 * beta)
        reg_loss = l1.sum() / max(1, num_fg) * a.get("reg_loss_weight", 1.0)
        total = cls_loss + reg_loss
        return {"total_loss": total, "cls_loss": cls_loss, "reg_loss": reg_loss}, {
            "labels": labels, "gt_offsets": gt_off, "anchors": anchors, "logits": logits, "offsets": offsets, "num_fg": num_fg}

    # ---- FreeAnchor.get_losses (models/det/free_anchor.py:20-142): RetinaNet's network, bag losses ------------
    def freeanchor_losses(self, batch):
        from . import freeanchor
        a = self.arch
        image = _t(box_ops.data_to_input(batch["data"], a["img_mean"], a["img_std"]))
        logits, offsets, sizes, _ = self.retinanet_forward(image)
        anchors = np.concatenate(box_ops.default_anchors(sizes, a["strides"], a["anchor_scales"], a["anchor_ratios"], a["anchor_offset"]), 0)
        num_valid = np.asarray(batch["im_info"])[:, 4].astype(np.int32)
        fa = a["freeanchor"]
        pos, neg = freeanchor.bag_losses(logits, offsets, anchors, np.asarray(batch["gt_boxes"], np.float32), num_valid,
                                         mean=fa["mean"], std=fa["std"], iou_thresh=fa["iou_thresh"], bucket=fa["bucket"],
                                         beta=a.get("smooth_l1_beta", 0.0), reg_weight=a.get("reg_loss_weight", 1.0),
                                         alpha=a.get("focal_alpha", 0.25), gamma=a.get("focal_gamma", 2.0))
        return {"total_loss": pos + neg, "pos_loss": pos, "neg_loss": neg}, {"anchors": anchors, "logits": logits, "offsets": offsets}

    # ---- FCOS (layers/head/point_head.py:137-151, models/det/fcos.py:114-179) ---------------------------------
    def point_head(self, feats):
        """PointHead.forward: conv -> GroupNorm(32) -> ReLU towers (Sequential indices 3i / 3i+1), cls_score, and
        offsets = relu(bbox_pred * scale_l) * stride_l, ctrness from the BOX tower (:143-144)."""
        nconv = self.arch.get("num_convs", 4)
        logits, offsets, ctrs = [], [], []
        for li, (f, stride) in enumerate(zip(feats, self.arch["strides"])):
            c, b = f, f
            for i in range(nconv):
                y = self._act(f"clsy{i}_{li}", self._conv(c, f"head.cls_subnet.{3 * i}", 1, 1), False)
                z = TF.group_norm(y, 32, self.p[f"head.cls_subnet.{3 * i + 1}.weight"], self.p[f"head.cls_subnet.{3 * i + 1}.bias"], 1e-5)
                c = self._act(f"cls{i}_{li}", z, True)
                y = self._act(f"boxy{i}_{li}", self._conv(b, f"head.bbox_subnet.{3 * i}", 1, 1), False)
                z = TF.group_norm(y, 32, self.p[f"head.bbox_subnet.{3 * i + 1}.weight"], self.p[f"head.bbox_subnet.{3 * i + 1}.bias"], 1e-5)
                b = self._act(f"box{i}_{li}", z, True)
            logits.append(self._act(f"logits_{li}", self._conv(c, "head.cls_score", 1, 1), False))
            raw = torch.cat([self._conv(b, "head.bbox_pred", 1, 1), self._conv(b, "head.ctrness", 1, 1)], dim=1)
            raw = self._act(f"raw_{li}", raw, False)
            offsets.append(self._q(TF.relu(raw[:, :4] * self.p["head.scales"][li]) * float(stride)))
            ctrs.append(raw[:, 4:5])
        return logits, offsets, ctrs

    @staticmethod
    def _giou_ltrb(p, t, eps=1e-8):
        """get_ltrb_boxes_iou(iou_type="giou") (layers/losses/iou_loss.py:9-56) on torch tensors."""
        b1 = torch.cat([-p[..., :2], p[..., 2:]], -1)
        b2 = torch.cat([-t[..., :2], t[..., 2:]], -1)
        a1 = (b1[..., 2] - b1[..., 0]).clamp(min=0) * (b1[..., 3] - b1[..., 1]).clamp(min=0)
        a2 = (b2[..., 2] - b2[..., 0]).clamp(min=0) * (b2[..., 3] - b2[..., 1]).clamp(min=0)
        wi = (torch.minimum(b1[..., 2], b2[..., 2]) - torch.maximum(b1[..., 0], b2[..., 0])).clamp(min=0)
        hi = (torch.minimum(b1[..., 3], b2[..., 3]) - torch.maximum(b1[..., 1], b2[..., 1])).clamp(min=0)
        ai = wi * hi
        au = a1 + a2 - ai
        iou = ai / au.clamp(min=eps)
        gw = torch.maximum(b1[..., 2], b2[..., 2]) - torch.minimum(b1[..., 0], b2[..., 0])
        gh = torch.maximum(b1[..., 3], b2[..., 3]) - torch.minimum(b1[..., 1], b2[..., 1])
        ac = gw * gh
        return iou - (ac - au) / ac.clamp(min=eps)

    def fcos_losses(self, batch):
        a = self.arch
        image = _t(box_ops.data_to_input(batch["data"], a["img_mean"], a["img_std"]))
        feats = self.fpn(self.backbone(image))
        logits, offsets, ctrs = self.point_head(feats)
        K = a["num_classes"]
        sizes = [tuple(f.shape[-2:]) for f in feats]
        logits = torch.cat([self._permute(x, K) for x in logits], dim=1).reshape(-1, K)
        offsets = torch.cat([self._permute(x, 4) for x in offsets], dim=1).reshape(-1, 4)
        ctrs = torch.cat([self._permute(x, 1) for x in ctrs], dim=1).reshape(-1)
        pts = box_ops.point_anchors(sizes, a["strides"], a["anchor_offset"], 1)
        num_valid = np.asarray(batch["im_info"])[:, 4].astype(np.int32)
        if a.get("atss"):                       # ATSS(FCOS): only the target assignment differs (models/det/atss.py)
            labels, gt_off, gt_ctr = box_ops.atss_ground_truth(pts, a["strides"], batch["gt_boxes"], num_valid, a["atss"]["scale"],
                                                               a["atss"]["topk"])
        else:
            labels, gt_off, gt_ctr = box_ops.fcos_ground_truth(pts, a["strides"], batch["gt_boxes"], num_valid, a["sizes_of_interest"],
                                                               a["center_sampling_radius"])
        labels_t = torch.from_numpy(labels.reshape(-1)).long()
        gt_off_t, gt_ctr_t = _t(gt_off.reshape(-1, 4)), _t(gt_ctr.reshape(-1))
        valid, fg = labels_t >= 0, labels_t > 0
        num_fg = float(fg.sum())
        sum_ctr = float(gt_ctr_t[fg].sum())
        tgt = torch.zeros_like(logits)
        tgt[fg, labels_t[fg] - 1] = 1
        x, t = logits[valid], tgt[valid]
        p = torch.sigmoid(x)
        ce = -(t * TF.logsigmoid(x) + (1 - t) * TF.logsigmoid(-x))
        al = a.get("focal_alpha", 0.25)
        fl = ce * (t * (1 - p) + (1 - t) * p) ** a.get("focal_gamma", 2.0) * (t * al + (1 - t) * (1 - al))
        cls_loss = fl.sum() / max(1.0, num_fg)
        giou = self._giou_ltrb(offsets[fg], gt_off_t[fg])
        reg_loss = ((1 - giou) * gt_ctr_t[fg]).sum() / max(1.0, sum_ctr) * a.get("reg_loss_weight", 1.0)
        xc, tc = ctrs[fg], gt_ctr_t[fg]
        ctr_loss = (-(tc * TF.logsigmoid(xc) + (1 - tc) * TF.logsigmoid(-xc))).sum() / max(1.0, num_fg)
        total = cls_loss + reg_loss + ctr_loss
        return {"total_loss": total, "cls_loss": cls_loss, "reg_loss": reg_loss, "ctr_loss": ctr_loss}, {
            "labels": labels, "gt_offsets": gt_off, "gt_ctr": gt_ctr, "logits": logits, "offsets": offsets, "num_fg": num_fg,
            "sum_ctr": sum_ctr}

    # ---- OTA (models/det/ota.py:62-233): the FCOS network (OTAPointHead with NORM_REG_TARGETS = PointHead's forward, the
    # centre-ness branch renamed ious_pred, point_head.py:154-212), dynamic top-k assignment, emd_losses -------------------
    def ota_losses(self, batch, forced=None):
        a = self.arch
        image = _t(box_ops.data_to_input(batch["data"], a["img_mean"], a["img_std"]))
        feats = self.fpn(self.backbone(image))
        logits, offsets, ious_pred = self.point_head(feats)
        K = a["num_classes"]
        sizes = [tuple(f.shape[-2:]) for f in feats]
        N = logits[0].shape[0]
        logits = torch.cat([self._permute(x, K) for x in logits], dim=1)
        offsets = torch.cat([self._permute(x, 4) for x in offsets], dim=1)
        ious_pred = torch.cat([self._permute(x, 1) for x in ious_pred], dim=1).reshape(-1)
        pts = box_ops.point_anchors(sizes, a["strides"], a["anchor_offset"], 1)
        num_valid = np.asarray(batch["im_info"])[:, 4].astype(np.int32)
        o = a["ota"]
        al, ga = a.get("focal_alpha", 0.25), a.get("focal_gamma", 2.0)
        if forced is None:
            labels, gt_off, gt_iou, aux = box_ops.ota_ground_truth(pts, a["strides"], logits.detach().numpy(), offsets.detach().numpy(),
                                                                   batch["gt_boxes"], num_valid, al, ga, o["reg_weight"],
                                                                   o["center_radius"], o["candidate_k"], o.get("matching", "topk"))
        else:                   # targets taken from the device path (gradient checks on identical assignments)
            labels, gt_off, gt_iou = forced
            aux = None
        logits, offsets = logits.reshape(-1, K), offsets.reshape(-1, 4)
        labels_t = torch.from_numpy(np.asarray(labels).reshape(-1)).long()
        gt_off_t, gt_iou_t = _t(np.asarray(gt_off).reshape(-1, 4)), _t(np.asarray(gt_iou).reshape(-1))
        fg = labels_t > 0
        num_fg = float(fg.sum())
        tgt = torch.zeros_like(logits)
        tgt[fg, labels_t[fg] - 1] = 1
        p = torch.sigmoid(logits)
        ce = -(tgt * TF.logsigmoid(logits) + (1 - tgt) * TF.logsigmoid(-logits))
        fl = ce * (tgt * (1 - p) + (1 - tgt) * p) ** ga * (tgt * al + (1 - tgt) * (1 - al))
        loss_cls = fl.sum() / max(1.0, num_fg)
        giou = self._giou_ltrb(offsets[fg], gt_off_t[fg])
        loss_box = (1 - giou).sum() / max(1.0, num_fg) * 2.0
        xi, ti = ious_pred[fg], gt_iou_t[fg]
        loss_iou = (-(ti * TF.logsigmoid(xi) + (1 - ti) * TF.logsigmoid(-xi))).sum() / max(1.0, num_fg) * 0.5
        total = loss_cls + loss_box + loss_iou
        return {"total_loss": total, "loss_cls": loss_cls, "loss_offsets": loss_box, "loss_ious": loss_iou}, {
            "labels": labels, "gt_offsets": gt_off, "gt_ious": gt_iou, "num_fg": num_fg, "aux": aux, "logits": logits, "offsets": offsets}

    # ---- Faster R-CNN (models/det/faster_rcnn.py:64-97, rpn.py:70-132, layers/head/rcnn.py:52-83) ---------------
    def rpn_head(self, feats):
        """rpn.py:78-100: per level relu(rpn_conv) -> cls (A) / offsets (4A)."""
        scores, offsets, A = [], [], None
        for li, f in enumerate(feats):
            t = self._act(f"rpn_t_{li}", self._conv(f, "rpn.rpn_conv", 1, 1), True)
            s = self._conv(t, "rpn.rpn_cls_score")
            o = self._conv(t, "rpn.rpn_bbox_offsets")
            A = s.shape[1]
            raw = self._act(f"rpn_raw_{li}", torch.cat([s, o], 1), False)      # the HIP path stores both in one fused row
            scores.append(raw[:, :A]); offsets.append(raw[:, A:])
        return scores, offsets

    def _linear(self, x, name):
        return TF.linear(x, self._q(self.p[name + ".weight"]), self.p[name + ".bias"])

    def roi_align_torch(self, feats, rois, batch_idx, strides, pool):
        """Differentiable roi_pool(..., "roi_align") built from oracle/rcnn_ops.py sample tables.
        feats[l]: (N, C, H, W).  Returns (R, C, PH, PW) like F.nn.roi_align."""
        from . import rcnn_ops
        PH, PW = pool
        lv = rcnn_ops.assign_roi_levels(rois, strides)
        C = feats[0].shape[1]
        outs = []
        for r in range(len(rois)):
            l, n = int(lv[r]), int(batch_idx[r])
            f = feats[l][n]                                   # (C, H, W)
            H, W = f.shape[1], f.shape[2]
            table = rcnn_ops.roi_align_sample_table(rois[r], 1.0 / strides[l], H, W, PH, PW, 2)
            idx, wts = [], []
            for pts in table:
                ii, ww = [], []
                for p in pts:
                    if p is None:
                        ii += [0, 0, 0, 0]; ww += [0.0, 0.0, 0.0, 0.0]
                    else:
                        y0, y1, x0, x1, w00, w01, w10, w11 = p
                        ii += [y0 * W + x0, y0 * W + x1, y1 * W + x0, y1 * W + x1]
                        ww += [float(w00), float(w01), float(w10), float(w11)]
                idx.append(ii); wts.append(ww)
            idx = torch.tensor(idx, dtype=torch.long)         # (PH*PW, 16)
            wts = torch.tensor(wts, dtype=torch.float32)
            g = f.reshape(C, H * W)[:, idx.reshape(-1)].reshape(C, PH * PW, -1)
            outs.append((g * wts[None]).sum(-1).reshape(C, PH, PW) * 0.25)
        return torch.stack(outs) if outs else torch.zeros((0, C, PH, PW))

    def faster_rcnn_losses(self, batch, keys, forced=None):
        """keys: dict rpn_pos / rpn_neg (N, A_total), rcnn_fg / rcnn_bg (N, post_k + Gmax) float32 in [0, 1).
        forced: optional dict(rois=list of (n_i, 4)) replacing the RPN proposals (to decouple the RCNN parity from the
        discrete proposal selection)."""
        from . import rcnn_ops
        a = self.arch
        image = _t(box_ops.data_to_input(batch["data"], a["img_mean"], a["img_std"]))
        feats = self.fpn(self.backbone(image))
        scores, offsets = self.rpn_head(feats)
        N = image.shape[0]
        sizes = [tuple(f.shape[-2:]) for f in feats]
        anchors_list = box_ops.default_anchors(sizes, a["strides"], a["anchor_scales"], a["anchor_ratios"], a["anchor_offset"])
        anchors = np.concatenate(anchors_list, 0)
        info = np.asarray(batch["im_info"], np.float32)
        num_valid = info[:, 4].astype(np.int32)
        rpn, rc = a["rpn"], a["rcnn"]
        A = scores[0].shape[1]
        # per level (N, HWA) / (N, HWA, 4) in the reference's (h, w, a) order (rpn.py:149-153)
        sc_l = [s.permute(0, 2, 3, 1).reshape(N, -1) for s in scores]
        of_l = [o.reshape(N, A, 4, o.shape[2], o.shape[3]).permute(0, 3, 4, 1, 2).reshape(N, -1, 4) for o in offsets]
        mean_r, std_r = a["rpn_box_reg"]
        rois_list = []
        for n in range(N):
            if forced is not None:
                rois_list.append(np.asarray(forced["rois"][n], np.float32))
                continue
            r, _, _ = rcnn_ops.rpn_proposals([s[n].detach().numpy() for s in sc_l], [o[n].detach().numpy() for o in of_l], anchors_list,
                                             info[n, :2], rpn["TRAIN_PREV_NMS_TOPK"], rpn["TRAIN_POST_NMS_TOPK"], rpn["NMS_THRESHOLD"],
                                             mean_r, std_r)
            rois_list.append(r)
        thr, labs, lq = a["matcher"]
        nsa = rpn["NUM_SAMPLE_ANCHORS"]
        rpn_labels, rpn_off = rcnn_ops.rpn_ground_truth(anchors, batch["gt_boxes"], num_valid, keys["rpn_pos"], keys["rpn_neg"], thr, labs,
                                                        lq, nsa, int(rpn["POSITIVE_ANCHOR_RATIO"] * nsa), mean_r, std_r)
        logits = torch.cat(sc_l, 1).reshape(-1)
        offs = torch.cat(of_l, 1).reshape(-1, 4)
        lab_t = torch.from_numpy(rpn_labels.reshape(-1)).long()
        valid, fg = lab_t >= 0, lab_t > 0
        nv = max(int(valid.sum()), 1)
        t = lab_t[valid].float()
        x = logits[valid]
        rpn_cls = (-(t * TF.logsigmoid(x) + (1 - t) * TF.logsigmoid(-x))).sum() / nv
        d = offs[fg] - _t(rpn_off.reshape(-1, 4))[fg]
        beta = a.get("rpn_beta", 0.0)
        l1 = d.abs() if beta < 1e-5 else torch.where(d.abs() < beta, 0.5 * d ** 2 / beta, d.abs() - 0.5 * beta)
        rpn_box = l1.sum() / nv
        # RCNN
        mean_c, std_c = a["rcnn_box_reg"]
        s_rois, s_labels, s_targets, s_bidx = [], [], [], []
        gtb = np.asarray(batch["gt_boxes"], np.float32)
        for n in range(N):
            rr, rl, rt = rcnn_ops.rcnn_ground_truth(rois_list[n], gtb[n, : num_valid[n]], keys["rcnn_fg"][n], keys["rcnn_bg"][n],
                                                    rc["NUM_ROIS"], rc["FG_RATIO"], rc["FG_THRESHOLD"], rc["BG_THRESHOLD_HIGH"],
                                                    rc["BG_THRESHOLD_LOW"], mean_c, std_c)
            s_rois.append(rr); s_labels.append(rl); s_targets.append(rt); s_bidx.append(np.full(len(rl), n))
        s_rois = np.concatenate(s_rois); s_labels = np.concatenate(s_labels); s_targets = np.concatenate(s_targets)
        s_bidx = np.concatenate(s_bidx)
        nl = len(rc["STRIDES"])
        pooled = self.roi_align_torch(feats[:nl], s_rois, s_bidx, list(rc["STRIDES"]), a["pool_size"])
        pooled = self._act("pooled", pooled.flatten(1), False)
        h1 = self._act("fc1", self._linear(pooled, "rcnn.fc1"), True)
        h2 = self._act("fc2", self._linear(h1, "rcnn.fc2"), True)
        raw = self._act("rcnn_raw", torch.cat([self._linear(h2, "rcnn.pred_cls"), self._linear(h2, "rcnn.pred_delta")], 1), False)
        K = a["num_classes"]
        cls_logits, deltas = raw[:, : K + 1], raw[:, K + 1:].reshape(-1, K, 4)
        R = max(len(s_labels), 1)
        lt = torch.from_numpy(s_labels).long()
        rcnn_cls = TF.cross_entropy(cls_logits, lt, reduction="sum") / R
        fgm = lt > 0
        d2 = deltas[fgm, lt[fgm] - 1] - _t(s_targets)[fgm]
        beta2 = a.get("rcnn_beta", 0.0)
        l2 = d2.abs() if beta2 < 1e-5 else torch.where(d2.abs() < beta2, 0.5 * d2 ** 2 / beta2, d2.abs() - 0.5 * beta2)
        rcnn_box = l2.sum() / R
        total = rpn_cls + rpn_box + rcnn_cls + rcnn_box
        return {"total_loss": total, "rpn_cls_loss": rpn_cls, "rpn_reg_loss": rpn_box, "rcnn_cls_loss": rcnn_cls,
                "rcnn_reg_loss": rcnn_box}, {"rois": rois_list, "s_rois": s_rois, "s_labels": s_labels, "s_targets": s_targets,
                                             "s_bidx": s_bidx, "rpn_labels": rpn_labels, "rpn_offsets": rpn_off}

    # ---- one training step: backward + SGD (solver/default_solver.py:96-124) --------------------
    def grads(self, loss):
        ps = [self.p[k] for k in self.trainable]
        gs = torch.autograd.grad(loss, ps, allow_unused=True)
        return {k: (g if g is not None else torch.zeros_like(p)) for k, g, p in zip(self.trainable, gs, ps)}

    def sgd_step(self, grads, state, lr, momentum=0.9, weight_decay=1e-4):
        with torch.no_grad():
            for k in self.trainable:
                w = self.p[k]
                g = grads[k] + weight_decay * w
                v = state.get(k)
                v = g.clone() if v is None else momentum * v + g
                state[k] = v
                w -= lr * v
        return state


def bilinear_up2(x):
    return TF.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
