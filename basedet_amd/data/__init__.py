"""Host side of the batch contract next to the hot path (basedet/data/collators/pad_collator.py, data/samplers/group_sampler.py).

Pure numpy, no MegEngine: the reference subclasses ``megengine.data.Collator`` / ``RandomSampler`` only for their interfaces.
What the training step consumes is the dict produced by ``DetectionPadCollator.apply`` -- ``data`` (N,3,H,W) float32,
``gt_boxes`` (N,G,5) float32, ``im_info`` (N,5) float32 -- which ``FPNDetector.pre_process`` then pads to a multiple of 32 and
normalises on the device (bd_pad_normalize)."""
from .collators import DetectionPadCollator, calculate_padding_shape  # noqa: F401
from .samplers import AspectRatioGroupSampler, GroupedRandomSampler  # noqa: F401
from .transforms import (Compose, RandomHorizontalFlip, ShortestEdgeResize, TestTimeCompose, ToMode,  # noqa: F401
                         build_transform)
