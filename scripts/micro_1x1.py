"""Same-box timing of the 1x1 convolutions of RetinaNet-R50 (generic kernel), forward / dgrad; BD_KNOB = bd_conv_set_patch3x3 mask."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
from basedet_amd import ops
if os.environ.get("BD_KNOB"):
    ops.L().bd_conv_set_patch3x3(int(os.environ["BD_KNOB"]))
for rep in range(2):
    for mode in sys.argv[1:] or ("fwd", "dgrad"):
        for (h, w, cin, cout) in ((200, 336, 64, 256), (200, 336, 256, 64), (100, 168, 128, 512), (100, 168, 512, 128), (50, 84, 256, 1024),
                                  (50, 84, 1024, 256), (25, 42, 512, 2048), (25, 42, 2048, 512), (100, 168, 512, 256), (50, 84, 1024, 512)):
            bench(16, h, w, cin, cout, R=1, pad=0, mode=mode)
        bench(16, 25, 42, 2048, 256, mode=mode, stride=2)
        bench(16, 50, 84, 512, 512, mode=mode, stride=2)
