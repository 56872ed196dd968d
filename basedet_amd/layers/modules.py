"""Module-level callables of basedet.layers (build_backbone / resnet50 / FPN, RetinaNetHead, PointHead) as NCHW-in / NCHW-out
wrappers over the HIP trunk and head kernels.

The reference composes a detector from these modules (models/det/retinanet.py:40-52, layers/backbone/build.py:20-39,
layers/head/retina_head.py:9-112, layers/head/point_head.py:12-151).  Here the detector owns its layers in one arena
(models/fpn_base.py), so each wrapper holds a detector instance and runs the matching slice of its forward: same inputs and outputs
as the reference module (lists / dicts of NCHW fp32 tensors), computed by the bf16 NHWC kernels.

The head modules also run BACKWARD (round 4): `head.backward(d_logits, d_offsets[, d_ctrness])` takes the gradients of the forward's outputs
(same lists of NCHW tensors), runs the detector's explicit dgrad / wgrad schedule for the head and returns the gradients with respect to
the input features; `head.grads()` then holds the parameter gradients under the reference's names (what MegEngine's GradManager would
hand over for `layers/head/retina_head.py` / `point_head.py`).  So a head can be composed with other code in a training loop; whole-model
training still goes through the model classes, which fuse target assignment and losses into the same schedule."""
import torch

from .. import ops


def _to_nchw(t, N, H, W, C=None):
    v = t.float().view(N, H, W, -1).permute(0, 3, 1, 2).contiguous()
    return v if C is None else v[:, :C].contiguous()


def _level(t, geom, i, C=None):
    N = geom.N
    v = t.view(N, geom.pix_per_img, -1)[:, geom.off[i]: geom.off[i] + geom.H[i] * geom.W[i]]
    return _to_nchw(v.contiguous(), N, geom.H[i], geom.W[i], C)


class _Trunk:
    def __init__(self, cfg, params=None, model=None):
        if model is None:
            from ..models import FCOS, FasterRCNN, RetinaNet
            kind = {"FCOS": FCOS, "ATSS": FCOS, "OTA": FCOS, "FasterRCNN": FasterRCNN}.get(cfg.MODEL.NAME, RetinaNet)
            model = kind(cfg, params=params)
        self.model = model
        self.cfg = cfg

    def _run_trunk(self, image):
        m = self.model
        pre = m.pre_process({"data": image})
        pl = pre["plan"]
        head_forward, m.head_forward = m.head_forward, (lambda _pl: None)
        try:
            m.network_forward(pl)
        finally:
            m.head_forward = head_forward
        return pl


class FPN(_Trunk):
    """layers/backbone/fpn_backbone.py:17-160 on top of the ResNet bottom-up (build_backbone): image (N, 3, H, W) -- raw pixel values,
    normalised inside like the models' pre_process -- -> {"p3": ..., ..., "p7": ...} NCHW fp32."""

    def __call__(self, image):
        pl = self._run_trunk(image)
        names = self.cfg.MODEL.FPN.OUT_FEATURES
        return {n: _level(pl.P, pl.pyr, i) for i, n in enumerate(names)}

    def output_shape(self):
        ch = self.cfg.MODEL.FPN.OUT_CHANNELS
        return {n: dict(channels=ch, stride=s) for n, s in zip(self.cfg.MODEL.FPN.OUT_FEATURES, self.cfg.MODEL.FPN.STRIDES)}


def build_backbone(cfg, params=None):
    """layers/backbone/build.py:20-39: the ResNet named by cfg.MODEL.BACKBONE.NAME under the FPN of cfg.MODEL.FPN."""
    return FPN(cfg, params=params)


class _ResNet(_Trunk):
    """models/cls/resnet.py:236-252 `extract_features`: {"res2".."res5"} of the frozen-BN ResNet, NCHW fp32."""

    def __call__(self, image):
        pl = self._run_trunk(image)
        m = self.model
        out = {}
        for i, blk in enumerate(m.blocks):
            last = i + 1 == len(m.blocks) or m.blocks[i + 1]["layer"] != blk["layer"]
            if last:
                b = pl.blk[i]
                out[f"res{blk['layer'] + 1}"] = _to_nchw(b.out, pl.N, b.gout.H[0], b.gout.W[0])
        return out

    extract_features = __call__


def _resnet(name):
    def make(cfg=None, params=None, **kw):
        from ..configs import RetinaNetConfig
        cfg = cfg or RetinaNetConfig()
        cfg.MODEL.BACKBONE.NAME = name
        if name in ("resnet18", "resnet34"):
            cfg.MODEL.BACKBONE.OUT_FEATURE_CHANNELS = [128, 256, 512]
            cfg.MODEL.FPN.TOP_BLOCK_IN_CHANNELS = 512
        return _ResNet(cfg, params=params)
    make.__name__ = name
    make.__doc__ = f"models/cls/resnet.py: {name} feature extractor (FrozenBN, stem + res2..res5)."
    return make


resnet18, resnet34, resnet50, resnet101 = (_resnet(n) for n in ("resnet18", "resnet34", "resnet50", "resnet101"))


class _Head(_Trunk):
    def _load(self, features):
        """NCHW pyramid features -> the detector's NHWC multi-level buffer."""
        m = self.model
        N, _, h0, w0 = features[0].shape
        s0 = m.strides[0]
        pl = m._plan(N, h0 * s0, w0 * s0)
        assert len(features) == pl.pyr.nlev, "input features expected {}, got {}".format(pl.pyr.nlev, len(features))
        P = pl.P.view(N, pl.pyr.pix_per_img, -1)
        for i, f in enumerate(features):
            assert tuple(f.shape[2:]) == (pl.pyr.H[i], pl.pyr.W[i]), "feature sizes must follow one padded image size"
            P[:, pl.pyr.off[i]: pl.pyr.off[i] + f.shape[2] * f.shape[3]] = \
                f.permute(0, 2, 3, 1).reshape(N, -1, f.shape[1]).to(torch.bfloat16)
        return pl


def _store(dst, geom, grads, C):
    """lists of NCHW gradients -> the detector's multi-level NHWC gradient buffer (channels beyond C -- padding columns -- zeroed)."""
    N = geom.N
    v = dst.view(N, geom.pix_per_img, -1)
    v.zero_()
    for i, g in enumerate(grads):
        assert tuple(g.shape[2:]) == (geom.H[i], geom.W[i]) and g.shape[1] == C, "gradient shapes must match the forward outputs"
        v[:, geom.off[i]: geom.off[i] + g.shape[2] * g.shape[3], :C] = g.permute(0, 2, 3, 1).reshape(N, -1, C).to(dst.dtype)


class _HeadBackward:
    def _finish(self, pl):
        m = self.model
        m._join_wgrads()
        torch.cuda.current_stream().synchronize()
        return [_level(pl.g_P, pl.pyr, i) for i in range(pl.pyr.nlev)]

    def grads(self):
        """Parameter gradients of the last backward() under the reference's names (`head.*`)."""
        return {k: v for k, v in self.model.reference_grads().items() if k.startswith("head.")}


class RetinaNetHead(_Head, _HeadBackward):
    """layers/head/retina_head.py:9-112: features [P3..P7] -> (logits [(N, A*K, H, W)], offsets [(N, A*4, H, W)])."""

    def __init__(self, cfg, input_shape=None, params=None, model=None):
        super().__init__(cfg, params=params, model=model)

    def __call__(self, features):
        pl = self._load(features)
        m = self.model
        m.head_forward(pl)
        n = pl.pyr.nlev
        self._pl = pl
        return ([_level(pl.logits, pl.pyr, i) for i in range(n)],
                [_level(pl.offsets, pl.pyr, i, m.num_anchors * 4) for i in range(n)])

    def backward(self, d_logits, d_offsets):
        """Gradients of the last forward's outputs -> gradients of its input features (list of NCHW fp32), parameter gradients in grads()."""
        pl, m = self._pl, self.model
        m._cur = pl
        m._begin_wgrads()
        _store(pl.d_logits, pl.pyr, d_logits, m.num_anchors * m.num_classes)
        _store(pl.d_offsets, pl.pyr, d_offsets, m.num_anchors * 4)
        m.head_backward(pl, pl.wgrad_ws, pl.colsum_ws)
        return self._finish(pl)


class PointHead(_Head, _HeadBackward):
    """layers/head/point_head.py:12-151: features -> (logits [(N, K, H, W)], offsets [(N, 4, H, W)], ctrness [(N, 1, H, W)])."""

    def __init__(self, cfg, input_shape=None, params=None, model=None):
        if model is None:
            from ..models import FCOS
            model = FCOS(cfg, params=params)
        super().__init__(cfg, model=model)

    def __call__(self, features):
        pl = self._load(features)
        self.model.head_forward(pl)
        n = pl.pyr.nlev
        logits = [_level(pl.logits, pl.pyr, i) for i in range(n)]
        offsets = [_level(pl.offsets, pl.pyr, i, 4) for i in range(n)]
        ctr = [_level(pl.raw, pl.pyr, i)[:, 4:5].contiguous() for i in range(n)]
        self._pl = pl
        return logits, offsets, ctr

    def backward(self, d_logits, d_offsets, d_ctrness):
        """As RetinaNetHead.backward; d_offsets are the gradients of the DECODED offsets (relu(x * scale_l) * stride_l, point_head.py:143)."""
        pl, m = self._pl, self.model
        m._cur = pl
        m._begin_wgrads()
        _store(pl.d_logits, pl.pyr, d_logits, m.num_classes)
        _store(pl.d_off, pl.pyr, d_offsets, 4)
        N = pl.pyr.N
        dc = pl.d_ctr.view(N, pl.pyr.pix_per_img)
        for i, g in enumerate(d_ctrness):
            dc[:, pl.pyr.off[i]: pl.pyr.off[i] + g.shape[2] * g.shape[3]] = g.reshape(N, -1).to(dc.dtype)
        m.head_backward(pl, pl.wgrad_ws, pl.colsum_ws)
        return self._finish(pl)
