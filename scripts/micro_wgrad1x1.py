"""Same-box timing of the 1x1 weight gradients of RetinaNet-R50 (BD_W1_PTR=1 in the environment: pointer staging instead of buffer loads)."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
for rep in range(2):
    for (h, w, cin, cout, s) in ((50, 84, 256, 1024, 1), (50, 84, 1024, 256, 1), (100, 168, 128, 512, 1), (100, 168, 512, 128, 1), (25, 42, 512, 2048, 1),
                                 (25, 42, 2048, 512, 1), (200, 336, 256, 128, 1), (100, 168, 512, 256, 1), (200, 336, 256, 512, 2), (100, 168, 512, 1024, 2)):
        bench(16, h, w, cin, cout, R=1, pad=0, mode="wgrad", stride=s)
