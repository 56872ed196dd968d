import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from micro_conv import bench
for rep in range(2):
    bench(16, 100, 168, 256, 256, mode="fwd")
    bench(16, 100, 168, 256, 256, mode="dgrad")
    bench(16, 50, 84, 256, 256, mode="fwd")
    bench(16, 25, 42, 512, 512, mode="fwd")
