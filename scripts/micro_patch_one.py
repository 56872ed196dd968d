"""One head-shape forward + dgrad of the 3x3 patch kernel (for rocprofv3 --pmc runs)."""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
sys.path.insert(0, _here)
from micro_conv import bench
bench(16, 100, 168, 256, 256, mode="fwd")
bench(16, 100, 168, 256, 256, mode="dgrad")
