"""Operator surface mirroring ``basedet.layers`` for the hot path (HIP kernels behind the same names)."""
from .box_ops import (  # noqa: F401
    DefaultAnchorGenerator, AnchorPointGenerator, Matcher, batched_nms, post_processing,
    post_process_with_empty_input, data_to_input, get_padded_tensor, permute_to_N_Any_K,
)
