"""A/B of the 3x3 patch-kernel variants on the head / backbone shapes (bd_conv_set_patch3x3: bit 3 = register-staged weights,
bit 4 = four-wave instance)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'scripts')
from basedet_amd import ops
from micro_conv import bench
names = {3: "dma8", 3 | 8: "regs8", 3 | 16: "w4"}
for knob in (3, 3 | 16, 3, 3 | 16):
    ops.L().bd_conv_set_patch3x3(knob)
    print("variant", names[knob], flush=True)
    for mode in ("fwd", "dgrad"):
        for (h, w, cin, cout) in ((100, 168, 256, 256), (100, 168, 128, 128), (50, 84, 256, 256), (25, 42, 512, 512), (100, 168, 256, 720)):
            bench(16, h, w, cin, cout, mode=mode)
