"""Evaluation next to the inference path (basedet/evaluators/coco_eval.py)."""
from .coco_eval import COCOEvaluator, bbox_eval  # noqa: F401
