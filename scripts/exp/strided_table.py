"""The strided launches of the RetinaNet-R50 step (first block of res3 / res4 / res5, P6 / P7), forward and data gradient, with their
algorithmic bytes: python scripts/exp/strided_table.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
from micro_conv import bench
for mode in ("fwd", "dgrad"):
    for (h, w, cin, cout, R) in ((200, 336, 128, 128, 3), (100, 168, 256, 256, 3), (50, 84, 512, 512, 3),
                                 (200, 336, 256, 512, 1), (100, 168, 512, 1024, 1), (50, 84, 1024, 2048, 1), (25, 42, 2048, 256, 3), (13, 21, 256, 256, 3)):
        bench(16, h, w, cin, cout, R=R, pad=R // 2, mode=mode, stride=2)
