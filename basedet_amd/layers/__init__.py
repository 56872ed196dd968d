"""Operator surface mirroring ``basedet.layers`` for the hot path (HIP kernels behind the same names)."""
from .box_ops import (  # noqa: F401
    DefaultAnchorGenerator, AnchorPointGenerator, Matcher, batched_nms, post_processing,
    post_process_with_empty_input, data_to_input, get_padded_tensor, get_multiple_size, permute_to_N_Any_K,
)
from .losses import (  # noqa: F401
    binary_cross_entropy, iou_loss, sigmoid_focal_loss, smooth_l1_loss, weighted_cross_entropy,
)
from .roi_pool import assign_rois, roi_pool, sample_labels  # noqa: F401
from .modules import FPN, PointHead, RetinaNetHead, build_backbone, resnet18, resnet34, resnet50, resnet101  # noqa: F401
from ..structures import box_center, box_ioa, box_iou  # noqa: F401
