"""Same-box timing of the 3x3 weight-gradient kernel (run once per library: BASEDET_HIP_LIB selects it)."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
for rep in range(2):
    for (h, w, cin, cout, s) in ((100, 168, 256, 256, 1), (50, 84, 256, 256, 1), (100, 168, 128, 128, 1), (25, 42, 512, 512, 1), (100, 168, 256, 720, 1),
                                 (200, 336, 128, 128, 2)):
        bench(16, h, w, cin, cout, mode="wgrad", stride=s)
