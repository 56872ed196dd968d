"""res3 / res4 3x3 stride-1 layers under the patch-kernel routes of bd_conv_desc.route[1]: 3 = default, 3|256 = conv3x3_pp128 for every shape it takes,
3|64 = no 256-channel staggered instance.   python scripts/exp/res3_3x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops
import micro_conv

for knob in (3, 3 | 256, 3 | 64):
    ops.set_route(patch3x3=knob)
    print(f"== bd_conv_desc.route[1]({knob})")
    for (H, W, C) in ((100, 168, 128), (50, 84, 256), (25, 42, 512)):
        for mode in ("fwd", "dgrad"):
            micro_conv.bench(16, H, W, C, C, mode=mode, iters=20)
            print("     ", ops.L().bd_conv_last_kernel().decode())
ops.set_route(patch3x3=3)
