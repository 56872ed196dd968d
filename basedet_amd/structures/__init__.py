"""``basedet.structures`` surface for the hot path: Boxes, BoxCoder, PointCoder, Container."""
import torch

from .. import ops


class Container(dict):
    """structures/container.py:5 (easydict-like)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class Boxes(torch.Tensor):
    """structures/boxes.py:10-26: a (N, 4) xyxy tensor subclass; pairwise ops run on the HIP kernels."""

    @staticmethod
    def __new__(cls, boxes):
        assert isinstance(boxes, torch.Tensor)
        assert boxes.ndim == 2
        assert boxes.shape[1] == 4
        return boxes.as_subclass(cls)

    def _raw(self):
        return self.as_subclass(torch.Tensor).float().contiguous()

    @property
    def centers(self):
        r = self._raw()
        return (r[:, :2] + r[:, 2:]) / 2.0

    @property
    def width(self):
        r = self._raw()
        return r[:, 2] - r[:, 0]

    @property
    def height(self):
        r = self._raw()
        return r[:, 3] - r[:, 1]

    @property
    def area(self):
        return self.width * self.height

    def iou(self, boxes):
        return ops.box_pairwise(self._raw(), torch.as_tensor(boxes).as_subclass(torch.Tensor).float().contiguous(), 0)

    def ioa(self, boxes):
        return ops.box_pairwise(self._raw(), torch.as_tensor(boxes).as_subclass(torch.Tensor).float().contiguous(), 1)

    def intersection(self, boxes):
        return ops.box_pairwise(self._raw(), torch.as_tensor(boxes).as_subclass(torch.Tensor).float().contiguous(), 2)

    def giou(self, boxes):
        return ops.box_pairwise(self._raw(), torch.as_tensor(boxes).as_subclass(torch.Tensor).float().contiguous(), 3)

    def scale(self, scale_ratios, inplace=True):
        if isinstance(scale_ratios, (int, float)):
            scale_ratios = (scale_ratios, scale_ratios)
        rh, rw = scale_ratios
        t = self if inplace else self.clone()
        r = t.as_subclass(torch.Tensor)
        r[:, 0::2] *= rw
        r[:, 1::2] *= rh
        return t

    def clip(self, sizes, inplace=True):
        if isinstance(sizes, (int, float)):
            sizes = (sizes, sizes)
        h, w = sizes
        t = self if inplace else self.clone()
        r = t.as_subclass(torch.Tensor)
        r[:, 0::2] = r[:, 0::2].clamp(0, w)
        r[:, 1::2] = r[:, 1::2].clamp(0, h)
        return t


class BoxCoder:
    """structures/boxcoder.py:30-98."""

    def __init__(self, reg_mean=(0.0, 0.0, 0.0, 0.0), reg_std=(1.0, 1.0, 1.0, 1.0)):
        self.reg_mean = [float(v) for v in reg_mean]
        self.reg_std = [float(v) for v in reg_std]

    def encode(self, bbox, gt):
        return ops.box_encode(bbox.float().contiguous(), gt.float().contiguous(), self.reg_mean, self.reg_std)

    def decode(self, anchors, deltas):
        assert deltas.shape[1] == 4, "HIP decode handles (A, 4) deltas"
        return ops.box_decode(anchors.float().contiguous(), deltas.float().contiguous(), self.reg_mean, self.reg_std)


class PointCoder:
    """structures/boxcoder.py:130-141."""

    def encode(self, point, gt):
        return torch.cat([point - gt[..., :2], gt[..., 2:] - point], dim=-1)

    def decode(self, anchors, deltas):
        return torch.stack([anchors[:, 0:1] - deltas[:, 0::4], anchors[:, 1:2] - deltas[:, 1::4],
                            anchors[:, 0:1] + deltas[:, 2::4], anchors[:, 1:2] + deltas[:, 3::4]], dim=2).reshape(deltas.shape)
