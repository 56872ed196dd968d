"""In-kernel clock of conv3x3_pp_kernel under sustained load (diagnostic build of the library with -DBD_PP_STAMP, path in BASEDET_HIP_LIB):
the RetinaNet head convolution launched back to back for ~3 s on random data, then d(s_memtime) / d(s_memrealtime) x 100 MHz of workgroup 0's
main loop and the launch's TFLOP/s."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops

N, C = 16, 256
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
geo = ops.Geom(N, [h for h, _ in sizes], [w for _, w in sizes])
d = ops.conv_desc(geo, geo, C, C, 3, 3, 1, 1)
x = torch.randn(geo.pixels, C, device="cuda").to(torch.bfloat16)
w = (torch.randn(C, 9, C, device="cuda") * 0.02).to(torch.bfloat16)
y = torch.empty(geo.pixels, C, device="cuda", dtype=torch.bfloat16)
bias = torch.zeros(C, device="cuda")
for _ in range(10):
    ops.conv2d_fwd(d, x, w, bias, y, flags=ops.EPI_RELU)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 9000
for i in range(n):
    if i == n - 100:
        s.record()
    ops.conv2d_fwd(d, x, w, bias, y, flags=ops.EPI_RELU)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 10.0
fl = 2.0 * geo.pixels * C * C * 9
st = (ctypes.c_ulonglong * 8)()
assert ops.L().bd_debug_pp_stamp(st) == 0
print(f"workgroup 300 (second round), cycles: setup + prologue {st[2]} (of which address setup before the first load {st[5]}), K loop {st[0]}, epilogue issue {st[3]}, store drain {st[4]}")
print(f"head conv fwd: {us:.1f} us per launch = {fl / us / 1e6:.0f} TFLOP/s after ~3 s of back-to-back launches; in-kernel clock {st[0] / st[1] * 100:.0f} MHz")
