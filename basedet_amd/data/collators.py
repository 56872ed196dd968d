"""DetectionPadCollator (basedet/data/collators/pad_collator.py:14-61)."""
from collections import defaultdict

import numpy as np

__all__ = ["DetectionPadCollator", "calculate_padding_shape"]


def calculate_padding_shape(original_shape, target_shape):
    """pad_collator.py:14-19: ((0, t - o), ...) per axis."""
    assert len(original_shape) == len(target_shape)
    return tuple((0, t - o) for o, t in zip(original_shape, target_shape))


class DetectionPadCollator:
    """Pads every field of the batch to the per-axis maximum (bottom / right, with ``pad_value``) and stacks it."""

    def __init__(self, pad_value: float = 0.0):
        self.pad_value = pad_value

    def apply(self, inputs):
        """inputs: iterable of (image (3,H,W), boxes (n,4), boxes_category (n,), info (orig_h, orig_w, ...)) -- pad_collator.py:32-61."""
        batch_data = defaultdict(list)
        for image, boxes, boxes_category, info in inputs:
            image = np.asarray(image)
            boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
            boxes_category = np.asarray(boxes_category).reshape(-1)
            batch_data["data"].append(image.astype(np.float32))
            batch_data["gt_boxes"].append(np.concatenate([boxes, boxes_category[:, np.newaxis]], axis=1).astype(np.float32))
            _, current_height, current_width = image.shape
            assert len(boxes) == len(boxes_category)
            num_instances = len(boxes)
            origin_height, origin_width = info[0], info[1]
            batch_data["im_info"].append(np.array([current_height, current_width, origin_height, origin_width, num_instances], dtype=np.float32))
        for key, value in batch_data.items():
            pad_shape = list(max(s) for s in zip(*[x.shape for x in value]))
            batch_data[key] = np.ascontiguousarray(
                [np.pad(v, calculate_padding_shape(v.shape, pad_shape), constant_values=self.pad_value) for v in value])
        return batch_data

    __call__ = apply
