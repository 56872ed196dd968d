"""Host-side augmentation next to the hot path: the reference's training pipeline (`configs/detection_cfg.py:42-53`:
ShortestEdgeResize(640..800, max 1333, "choice") -> RandomHorizontalFlip(0.5) -> ToMode("CHW"), composed over
("image", "boxes", "boxes_category")) and its test-time pipeline (`data/transforms/transforms.py:91-123` TestTimeCompose).

The reference takes ShortestEdgeResize / RandomHorizontalFlip / Compose from `megengine.data.transform` (third party, not under
/root/reference; they call cv2.resize with INTER_LINEAR).  This is a numpy restatement of their published behaviour -- pixel
values of the resize are cv2-style bilinear (half-pixel centres, edge clamp); parity with cv2's fixed-point arithmetic is
unpinned (cv2 is not installed here).  Pure numpy, no device work: what leaves this module is the sample the collator pads."""
import numpy as np

__all__ = ["ShortestEdgeResize", "RandomHorizontalFlip", "ToMode", "Compose", "TestTimeCompose", "build_transform"]


def resize_bilinear(img, out_h, out_w):
    """cv2.INTER_LINEAR semantics: source coordinate (dst + 0.5) * scale - 0.5, clamped to the image."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.copy()
    src = img.astype(np.float32)
    ys = (np.arange(out_h, dtype=np.float64) + 0.5) * (h / out_h) - 0.5
    xs = (np.arange(out_w, dtype=np.float64) + 0.5) * (w / out_w) - 0.5
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0).astype(np.float32); fx = (xs - x0).astype(np.float32)
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    fy = fy.reshape(-1, *([1] * (src.ndim - 1)))
    fxs = fx.reshape(1, -1, *([1] * (src.ndim - 2)))
    top = src[y0c][:, x0c] * (1 - fxs) + src[y0c][:, x1c] * fxs
    bot = src[y1c][:, x0c] * (1 - fxs) + src[y1c][:, x1c] * fxs
    out = top * (1 - fy) + bot * fy
    if np.issubdtype(img.dtype, np.integer):
        out = np.clip(np.rint(out), np.iinfo(img.dtype).min, np.iinfo(img.dtype).max)
    return out.astype(img.dtype)


class _Transform:
    """apply(sample_tuple) dispatches per field name of `order` (megengine VisionTransform protocol)."""
    order = ("image",)

    def apply(self, sample):
        if not isinstance(sample, tuple):
            return self._apply_image(sample)
        self._get_params(sample[self.order.index("image")])
        return tuple(getattr(self, "_apply_" + name)(v) for name, v in zip(self.order, sample))

    def _get_params(self, image):
        pass

    def _apply_image(self, image):
        return image

    def _apply_boxes(self, boxes):
        return boxes

    def _apply_boxes_category(self, c):
        return c


class ShortestEdgeResize(_Transform):
    """Scale so the short edge is `min_size` (one value drawn per sample: "choice" from the list, "range" between its two ends)
    unless the long edge would exceed `max_size` (then the long edge is max_size); sizes rounded half up as in megengine."""

    def __init__(self, min_size, max_size, sample_style="range", rng=None):
        self.min_size = (min_size, min_size) if isinstance(min_size, int) else tuple(min_size)
        self.max_size, self.sample_style = max_size, sample_style
        assert sample_style in ("range", "choice")
        self.rng = rng if rng is not None else np.random.default_rng()
        self._shape_info = None

    def _get_params(self, image):
        h, w = image.shape[:2]
        if self.sample_style == "range":
            size = int(self.rng.integers(self.min_size[0], self.min_size[1] + 1))
        else:
            size = int(self.min_size[self.rng.integers(len(self.min_size))])
        scale = size / min(h, w)
        nh, nw = (size, scale * w) if h < w else (scale * h, size)
        if max(nh, nw) > self.max_size:
            s = self.max_size / max(nh, nw)
            nh, nw = nh * s, nw * s
        self._shape_info = (h, w, int(nh + 0.5), int(nw + 0.5))

    def _apply_image(self, image):
        if self._shape_info is None or self._shape_info[:2] != image.shape[:2]:
            self._get_params(image)
        _, _, nh, nw = self._shape_info
        return resize_bilinear(image, nh, nw)

    def _apply_boxes(self, boxes):
        h, w, nh, nw = self._shape_info
        out = np.asarray(boxes, dtype=np.float32).copy()
        out[:, 0::2] *= nw / w
        out[:, 1::2] *= nh / h
        return out


class RandomHorizontalFlip(_Transform):
    def __init__(self, prob=0.5, rng=None):
        self.prob = prob
        self.rng = rng if rng is not None else np.random.default_rng()
        self._flip, self._w = False, 0

    def _get_params(self, image):
        self._flip = bool(self.rng.random() < self.prob)
        self._w = image.shape[1]

    def _apply_image(self, image):
        return image[:, ::-1].copy() if self._flip else image

    def _apply_boxes(self, boxes):
        out = np.asarray(boxes, dtype=np.float32).copy()
        if self._flip:            # x' = W - x, the two corners swap (megengine HorizontalFlip._apply_coords)
            out[:, 0], out[:, 2] = self._w - boxes[:, 2], self._w - boxes[:, 0]
        return out


class ToMode(_Transform):
    """HWC -> CHW (dtype kept) or NCHW (float32, batch axis added) (`data/transforms/transforms.py:56-88`)."""

    def __init__(self, mode="CHW"):
        assert mode in ("CHW", "NCHW"), f"unsupported mode: {mode}"
        self.mode = mode

    def _apply_image(self, image):
        if self.mode == "CHW":
            return np.ascontiguousarray(image.transpose(2, 0, 1))
        return np.ascontiguousarray(image.transpose(2, 0, 1)[None], dtype=np.float32)


class Compose:
    def __init__(self, transforms, order=("image", "boxes", "boxes_category")):
        self.transforms, self.order = list(transforms), tuple(order)
        for t in self.transforms:
            t.order = self.order

    def apply(self, sample):
        for t in self.transforms:
            sample = t.apply(sample)
        return sample

    __call__ = apply


class TestTimeCompose(Compose):
    """Image-only pipeline that also returns im_info = [resized H, resized W, original H, original W]
    (`data/transforms/transforms.py:98-113`)."""
    __test__ = False          # not a pytest class

    def __call__(self, image):
        oh, ow = image.shape[:2]
        shape = None
        for t in self.transforms:
            image = t.apply(image)
            if isinstance(t, ShortestEdgeResize):
                shape = t._shape_info[2:]
        if shape is None:
            shape = image.shape[-2:]
        return image, np.array([(*shape, oh, ow)], dtype=np.float32)


_BY_NAME = {"MGE_ShortestEdgeResize": ShortestEdgeResize, "MGE_RandomHorizontalFlip": RandomHorizontalFlip, "MGE_ToMode": ToMode,
            "ShortestEdgeResize": ShortestEdgeResize, "RandomHorizontalFlip": RandomHorizontalFlip, "ToMode": ToMode}

TRAIN_AUG = (("MGE_ShortestEdgeResize", dict(min_size=(640, 672, 704, 736, 768, 800), max_size=1333, sample_style="choice")),
             ("MGE_RandomHorizontalFlip", dict(prob=0.5)), ("MGE_ToMode", dict(mode="CHW")))
TEST_AUG = (("MGE_ShortestEdgeResize", dict(min_size=800, max_size=1333, sample_style="choice")),      # configs/extra_cfg.py:119-131
            ("ToMode", dict(mode="NCHW")))


def build_transform(spec=None, mode="train", rng=None):
    """`data/build.py` build_transform: a (name, kwargs) list -> Compose (train) / TestTimeCompose (test)."""
    spec = spec if spec is not None else (TRAIN_AUG if mode == "train" else TEST_AUG)
    ts = []
    for name, kw in spec:
        cls = _BY_NAME[name]
        kw = dict(kw)
        if cls in (ShortestEdgeResize, RandomHorizontalFlip):
            kw["rng"] = rng
        ts.append(cls(**kw))
    return Compose(ts) if mode == "train" else TestTimeCompose(ts, order=("image",))
