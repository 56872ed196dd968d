"""Same-box timing of the stride-2 convolutions of RetinaNet-R50, forward / dgrad (run once per library: BASEDET_HIP_LIB selects it)."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
from basedet_amd import ops
if os.environ.get('BD_KNOB'):
    ops.set_route(patch3x3=int(os.environ['BD_KNOB']))
for rep in range(2):
    for mode in sys.argv[1:] or ("dgrad",):
        for (h, w, cin, cout) in ((200, 336, 128, 128), (100, 168, 256, 256), (50, 84, 512, 512), (25, 42, 2048, 256)):
            bench(16, h, w, cin, cout, mode=mode, stride=2)
