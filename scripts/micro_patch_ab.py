"""Same-box timing of the 3x3 patch kernel on the head / backbone shapes (run once per library: BASEDET_HIP_LIB selects it)."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
for rep in range(2):
    for mode in ("fwd", "dgrad"):
        for (h, w, cin, cout) in ((100, 168, 256, 256), (100, 168, 128, 128), (50, 84, 256, 256), (25, 42, 512, 512)):
            bench(16, h, w, cin, cout, mode=mode)
