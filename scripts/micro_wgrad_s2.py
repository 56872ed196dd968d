"""Same-box timing of the stride-2 3x3 weight gradients of RetinaNet-R50 (run once per library: BASEDET_HIP_LIB selects it)."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
for rep in range(2):
    for (h, w, cin, cout) in ((200, 336, 128, 128), (100, 168, 256, 256), (50, 84, 512, 512), (25, 42, 2048, 256), (13, 21, 256, 256)):
        bench(16, h, w, cin, cout, mode="wgrad", stride=2)
    for (h, w, cin, cout) in ((200, 336, 256, 512), (100, 168, 512, 1024), (50, 84, 1024, 2048)):
        bench(16, h, w, cin, cout, R=1, pad=0, mode="wgrad", stride=2)
