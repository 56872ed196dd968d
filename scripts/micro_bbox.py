"""bbox_pred (256 -> 40) forward / dgrad through the two patch instances."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from basedet_amd import ops
from micro_conv import bench
for knob in (3 | 64, 3, 3 | 64, 3):
    ops.L().bd_conv_set_patch3x3(knob)
    print("knob", knob, flush=True)
    bench(16, 100, 168, 256, 40, mode="dgrad")
    bench(16, 100, 168, 256, 720, mode="dgrad")
