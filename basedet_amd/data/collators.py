"""The batch contract of the hot path on the host side: a list of per-image samples -> the three arrays the training step consumes.

Contract (what basedet's DetectionPadCollator hands to the models, pad_collator.py:38-49):
  data      (N, 3, Hmax, Wmax) float32 -- every image top-left aligned, padded bottom / right with ``pad_value``
  gt_boxes  (N, Gmax, 5)       float32 -- rows [x1, y1, x2, y2, category], padded with ``pad_value`` rows
  im_info   (N, 5)             float32 -- [H, W, original H, original W, number of boxes] of each image
The batch arrays are allocated once at their final shapes and the samples are copied into them (one pass over the inputs for the
shapes, one for the copies); nothing is padded per sample."""
import numpy as np

__all__ = ["DetectionPadCollator", "calculate_padding_shape"]


def calculate_padding_shape(original_shape, target_shape):
    """Per-axis (before, after) pad widths that grow ``original_shape`` to ``target_shape`` at the end of every axis."""
    assert len(original_shape) == len(target_shape)
    return tuple((0, int(t) - int(o)) for o, t in zip(original_shape, target_shape))


class DetectionPadCollator:
    def __init__(self, pad_value: float = 0.0):
        self.pad_value = pad_value

    def apply(self, inputs):
        """inputs: iterable of (image (3, H, W), boxes (n, 4), boxes_category (n,), info) with info[0:2] = original (H, W)."""
        samples = []
        for image, boxes, category, info in inputs:
            image = np.asarray(image)
            boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
            category = np.asarray(category, np.float32).reshape(-1)
            assert len(boxes) == len(category)
            samples.append((image, boxes, category, info))
        n = len(samples)
        chans = samples[0][0].shape[0] if n else 3
        hmax = max((s[0].shape[1] for s in samples), default=0)
        wmax = max((s[0].shape[2] for s in samples), default=0)
        gmax = max((len(s[1]) for s in samples), default=0)
        data = np.full((n, chans, hmax, wmax), self.pad_value, np.float32)
        gt_boxes = np.full((n, gmax, 5), self.pad_value, np.float32)
        im_info = np.empty((n, 5), np.float32)
        for i, (image, boxes, category, info) in enumerate(samples):
            _, h, w = image.shape
            g = len(boxes)
            data[i, :, :h, :w] = image
            gt_boxes[i, :g, :4] = boxes
            gt_boxes[i, :g, 4] = category
            im_info[i] = (h, w, info[0], info[1], g)
        return {"data": data, "gt_boxes": gt_boxes, "im_info": im_info}

    __call__ = apply
