"""Faster R-CNN bench with the deterministic (gather) RoIAlign backward: python scripts/frcnn_det.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basedet_amd.models.faster_rcnn as F
_orig = F.FasterRCNN._build_head
def _patched(self, add, params):
    _orig(self, add, params)
    self.deterministic_roi_bwd = True
F.FasterRCNN._build_head = _patched
sys.argv = ["bench.py", "--workload", "faster_rcnn_r50_800x1344", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-roofline"]
import bench
bench.main()
