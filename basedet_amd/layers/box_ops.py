"""basedet.layers.common operators on the HIP kernels (same names / argument meaning / asserts as the reference)."""
import math
from typing import Optional

import numpy as np
import torch

from .. import ops
from ..structures import Boxes, Container


class DefaultAnchorGenerator:
    """layers/common/anchor_generator.py:52-122."""

    def __init__(self, anchor_scales=[[32], [64], [128], [256], [512]], anchor_ratios=[[0.5, 1, 2]],
                 strides=[4, 8, 16, 32, 64], offset=0):
        self.anchor_scales = np.array(anchor_scales, dtype=np.float32)
        self.anchor_ratios = np.array(anchor_ratios, dtype=np.float32)
        self.strides = strides
        self.offset = offset
        self.num_features = len(strides)
        self._base = None

    @property
    def anchor_dim(self):
        return 4

    @staticmethod
    def generate_base_anchors(scales, ratios):
        base = []
        for area in [s ** 2.0 for s in scales]:
            for ratio in ratios:
                w = math.sqrt(area / ratio)
                h = ratio * w
                base.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return base

    def base_anchors(self, device):
        if self._base is None:
            scales, ratios = self.anchor_scales.tolist(), self.anchor_ratios.tolist()
            if len(scales) == 1:
                scales *= self.num_features
            if len(ratios) == 1:
                ratios *= self.num_features
            assert len(scales) == self.num_features and len(ratios) == self.num_features
            self._base = [torch.tensor(self.generate_base_anchors(s, r), dtype=torch.float32, device=device)
                          for s, r in zip(scales, ratios)]
        return self._base

    def generate_anchors_by_features(self, sizes, device):
        assert len(sizes) == self.num_features, "input features expected {}, got {}".format(self.num_features, len(sizes))
        out = []
        for (h, w), stride, base in zip(sizes, self.strides, self.base_anchors(device)):
            t = torch.empty((h * w * base.shape[0], 4), dtype=torch.float32, device=device)
            out.append(ops.anchors_generate(h, w, stride, self.offset, base, t))
        return out

    def __call__(self, featmaps):
        return self.generate_anchors_by_features([tuple(f.shape[-2:]) for f in featmaps], featmaps[0].device)


class AnchorPointGenerator:
    """layers/common/anchor_generator.py:125-165."""

    def __init__(self, num_anchors=1, strides=(4, 8, 16, 32, 64), offset=0.5):
        self.num_anchors, self.strides, self.offset = num_anchors, strides, offset
        self.num_features = len(strides)

    @property
    def anchor_dim(self):
        return 2

    def generate_anchors_by_features(self, sizes, device):
        assert len(sizes) == self.num_features, "input features expected {}, got {}".format(self.num_features, len(sizes))
        out = []
        for (h, w), stride in zip(sizes, self.strides):
            t = torch.empty((h * w * self.num_anchors, 2), dtype=torch.float32, device=device)
            out.append(ops.points_generate(h, w, stride, self.offset, self.num_anchors, t))
        return out

    def __call__(self, featmaps):
        return self.generate_anchors_by_features([tuple(f.shape[-2:]) for f in featmaps], featmaps[0].device)


class Matcher:
    """layers/common/matcher.py:19-51.  `matcher(matrix)` with a (G, A) similarity matrix -> (match_indices, labels), any number of
    thresholds / labels (bd_matcher_matrix).  `match(gt_boxes, anchors)` is the fused form the training step uses: IoU, matching
    and labelling without materialising the matrix (bd_retina_assign_encode; two thresholds, labels [0, -1, 1])."""

    def __init__(self, thresholds, labels, allow_low_quality_matches=False):
        assert len(thresholds) + 1 == len(labels), "thresholds and labels are not matched"
        assert all(low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:]))
        thresholds = list(thresholds)                     # the reference mutates the caller's list in place (matcher.py:24-25)
        self.inner_thresholds = thresholds
        self.thresholds = [-float("inf")] + thresholds + [float("inf")]
        self.labels = list(labels)
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, matrix):
        assert len(matrix.shape) == 2
        m = matrix.float().contiguous()
        G, A = m.shape
        dev = m.device
        idx = torch.empty((A,), dtype=torch.int32, device=dev)
        labels = torch.empty((A,), dtype=torch.int32, device=dev)
        ws = torch.empty((max(G, 1),), dtype=torch.float32, device=dev)
        from .._lib import check, f32arr, i32arr, ptr, stream_ptr
        check(ops.L().bd_matcher_matrix(ptr(m), G, A, f32arr(self.inner_thresholds), i32arr(self.labels), len(self.inner_thresholds),
                                        int(self.allow_low_quality_matches), ptr(idx), ptr(labels), ptr(ws), stream_ptr()),
              "bd_matcher_matrix")
        return idx, labels

    def match(self, gt_boxes_with_labels, anchors):
        """gt (G,5) with class in column 4 -> (match_indices, labels) with labels in {-1, 0, class}."""
        assert list(self.labels) == [0, -1, 1] and len(self.inner_thresholds) == 2, "the fused matcher supports labels [0, -1, 1]"
        dev = anchors.device
        gt = gt_boxes_with_labels.reshape(1, -1, 5).contiguous().float()
        G = gt.shape[1]
        A = anchors.shape[0]
        labels = torch.empty((1, A), dtype=torch.int32, device=dev)
        idx = torch.empty((1, A), dtype=torch.int32, device=dev)
        offs = torch.empty((1, A, 4), dtype=torch.float32, device=dev)
        nfg = torch.zeros((1,), dtype=torch.int32, device=dev)
        ws = torch.empty((max(G, 1),), dtype=torch.float32, device=dev)
        ng = torch.tensor([G], dtype=torch.int32, device=dev)
        if G == 0:
            gt = torch.zeros((1, 1, 5), dtype=torch.float32, device=dev)
        ops.retina_assign_encode(anchors, gt, ng, self.inner_thresholds[0], self.inner_thresholds[1], self.allow_low_quality_matches,
                                 (0, 0, 0, 0), (1, 1, 1, 1), labels, idx, offs, nfg, ws)
        return idx[0], labels[0]


def batched_nms(boxes, scores, idxs, iou_thresh: float, max_output: Optional[int] = None):
    """layers/common/post_processing.py:17-47."""
    assert boxes.ndim == 2 and boxes.shape[1] == 4, "the expected shape of boxes is (N, 4)"
    assert scores.ndim == 1, "the expected shape of scores is (N,)"
    assert idxs.ndim == 1, "the expected shape of idxs is (N,)"
    assert boxes.shape[0] == scores.shape[0] == idxs.shape[0], "number of boxes, scores and idxs are not matched"
    return ops.batched_nms(boxes.float().contiguous(), scores.float().contiguous(), idxs.to(torch.int32).contiguous(),
                           iou_thresh, max_output)


def post_processing(boxes_container, img_info, iou_threshold, process_method="nms", max_detections_per_image=None):
    """layers/common/post_processing.py:78-103: NMS, rescale to the original image, clip."""
    keep = batched_nms(boxes_container.boxes, boxes_container.box_scores, boxes_container.box_labels,
                       iou_thresh=iou_threshold, max_output=max_detections_per_image).long()
    kept = Container(boxes=Boxes(boxes_container.boxes[keep]), box_scores=boxes_container.box_scores[keep],
                     box_labels=boxes_container.box_labels[keep])
    info = img_info.float().cpu()
    ratios = (float(info[0, 2] / info[0, 0]), float(info[0, 3] / info[0, 1]))
    kept.boxes.scale(ratios).clip((float(info[0, 2]), float(info[0, 3])))
    return kept


def post_process_with_empty_input(boxes, box_scores, box_labels, img_info, iou_threshold=0.5, max_detections_per_image=100):
    """layers/common/post_processing.py:50-75."""
    if not boxes:
        e = torch.zeros((0,))
        return Container(boxes=e, box_scores=e, box_labels=e)
    c = Container(boxes=Boxes(torch.cat(boxes, 0)), box_scores=torch.cat(box_scores, 0), box_labels=torch.cat(box_labels, 0))
    return post_processing(c, img_info, iou_threshold=iou_threshold, max_detections_per_image=max_detections_per_image)


def get_multiple_size(n, multiple=32):
    return (n + multiple - 1) // multiple * multiple


def get_padded_tensor(tensor, multiple_number=32, pad_value=0):
    """layers/common/pre_processing.py:26-49 (layout op; the training path fuses it into bd_pad_normalize)."""
    *size, h, w = tensor.shape
    out = torch.full((*size, get_multiple_size(h, multiple_number), get_multiple_size(w, multiple_number)), pad_value,
                     dtype=tensor.dtype, device=tensor.device)
    out[..., :h, :w] = tensor
    return out


def data_to_input(image, mean=None, std=None):
    """layers/common/pre_processing.py:11-19 -> fp32 NCHW via bd_pad_normalize_nchw."""
    image = torch.as_tensor(image).float().cuda().contiguous()
    n, c, h, w = image.shape
    assert c == 3
    hp, wp = get_multiple_size(h), get_multiple_size(w)
    out = torch.empty((n, 3, hp, wp), dtype=torch.float32, device=image.device)
    mean = [0, 0, 0] if mean is None else [float(v) for v in torch.as_tensor(mean).flatten()]
    std = [1, 1, 1] if std is None else [float(v) for v in torch.as_tensor(std).flatten()]
    return ops.pad_normalize_nchw(image, hp, wp, mean, std, out)


def permute_to_N_Any_K(tensor, K):
    """layers/common/function.py:26-32.  NHWC kernels already produce this layout; kept for NCHW callers."""
    assert tensor.ndim == 4
    return tensor.permute(0, 2, 3, 1).reshape(tensor.shape[0], -1, K)
