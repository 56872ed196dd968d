"""A/B of the 3x3 patch-kernel variants on the head / backbone shapes (bd_conv_set_patch3x3 bit 3: 1 = register-staged)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'scripts')
from basedet_amd import ops
from micro_conv import bench
for knob, name in ((3 | 8, "regs"), (3, "dma"), (3 | 8, "regs"), (3, "dma")):
    ops.L().bd_conv_set_patch3x3(knob)
    print("variant", name, flush=True)
    for mode in ("fwd", "dgrad"):
        for (h, w, cin, cout) in ((100, 168, 256, 256), (100, 168, 128, 128), (50, 84, 256, 256), (25, 42, 512, 512), (100, 168, 256, 720)):
            bench(16, h, w, cin, cout, mode=mode)
