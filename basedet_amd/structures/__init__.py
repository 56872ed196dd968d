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

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # results of tensor ops are plain tensors; only (N, 4) indexing results come back as Boxes (boxes.py:214-219)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    def __getitem__(self, idx):
        """structures/boxes.py:214-219: a (N, 4) result stays a Boxes, anything else is a plain tensor."""
        out = self.as_subclass(torch.Tensor)[idx]
        if out.ndim == 2 and out.shape[1] == 4:
            out = Boxes(out)
        return out

    def numpy(self):
        return self.as_subclass(torch.Tensor).detach().cpu().numpy()

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
        """structures/boxes.py:190-212: scale_ratios = (height, width) factors, or one number for both."""
        if torch.is_tensor(scale_ratios):
            scale_ratios = scale_ratios.tolist()
        if isinstance(scale_ratios, (int, float)):
            scale_ratios = (scale_ratios, scale_ratios)
        assert len(scale_ratios) == 2
        rh, rw = scale_ratios
        r = self.as_subclass(torch.Tensor)
        if not inplace:
            r = r.clone()
        r[:, 0::2] *= rw
        r[:, 1::2] *= rh
        return self if inplace else Boxes(r)

    def clip(self, sizes, inplace=True):
        """structures/boxes.py:150-177: sizes = (height, width) of the image region."""
        if torch.is_tensor(sizes):
            sizes = sizes.tolist()
        if isinstance(sizes, (int, float)):
            sizes = (sizes, sizes)
        assert len(sizes) == 2
        h, w = sizes
        r = self.as_subclass(torch.Tensor)
        if not inplace:
            r = r.clone()
        r[:, 0::2] = r[:, 0::2].clamp(0, w)
        r[:, 1::2] = r[:, 1::2].clamp(0, h)
        return self if inplace else Boxes(r)

    def filter_by_size(self, sizes=0):
        """structures/boxes.py:132-148: boolean keep mask of the boxes larger than `sizes`.  The reference binds `h, w = self.width,
        self.height` (sic) and then tests `w > sizes[0]`, `h > sizes[1]`: sizes[0] is compared with the box HEIGHT and sizes[1] with
        its WIDTH, i.e. sizes = (height, width) as documented; kept."""
        if torch.is_tensor(sizes):
            sizes = sizes.tolist()
        if isinstance(sizes, (int, float)):
            sizes = (sizes, sizes)
        assert len(sizes) == 2
        return (self.height > sizes[0]) & (self.width > sizes[1])

    def cat(self, boxes, inplace=True):
        """structures/boxes.py:179-188: concatenated boxes.  The reference's in-place branch assigns a longer tensor into `self[:]`,
        which cannot succeed; both branches return the concatenation here."""
        return Boxes(torch.cat([self.as_subclass(torch.Tensor), torch.as_tensor(boxes).as_subclass(torch.Tensor)], 0))


def box_iou(boxes1, boxes2):
    """structures/op_patch.py:81-97."""
    return Boxes(torch.as_tensor(boxes1).as_subclass(torch.Tensor)).iou(boxes2)


def box_ioa(boxes1, boxes2):
    """structures/op_patch.py:211-227."""
    return Boxes(torch.as_tensor(boxes1).as_subclass(torch.Tensor)).ioa(boxes2)


def box_center(boxes):
    """structures/op_patch.py:101-130."""
    return Boxes(torch.as_tensor(boxes).as_subclass(torch.Tensor)).centers


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
