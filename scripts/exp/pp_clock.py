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
if len(sys.argv) > 1:
    ops.set_route(patch3x3=int(sys.argv[1]))
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
st = (ctypes.c_ulonglong * 64)()
assert ops.L().bd_debug_pp_stamp(st) == 0
v = list(st)
clk = (v[61] - v[0]) / max(1, v[63] - v[62]) * 100
print(f"head conv fwd: {us:.1f} us per launch = {fl / us / 1e6:.0f} TFLOP/s after ~3 s of back-to-back launches; in-kernel clock {clk:.0f} MHz")
print(f"workgroup 100: decode {v[1] - v[0]}, prologue {v[2] - v[1]} cycles")
i, t = 2, 0
while i + 5 < 61 and v[i + 5] > v[i] > 0:
    nxt = v[i + 6] - v[i + 5] if (i + 6 < 61 and v[i + 6] > v[i + 5]) else 0
    print(f"tile {t}: K loop {v[i + 2] - v[i]} (last K block incl. next decode {v[i + 2] - v[i + 1]}), epilogue: to first group converted {v[i + 3] - v[i + 2]}, "
          f"wait {v[i + 4] - v[i + 3]}, rest + stores {v[i + 5] - v[i + 4]}; to next K loop {nxt}")
    i += 6; t += 1
print(f"whole workgroup {v[61] - v[0]} cycles")

sp = (ctypes.c_ulonglong * 2048)()
lib = ops.L()
lib.bd_debug_pp_span.argtypes = [ctypes.c_void_p]
assert lib.bd_debug_pp_span(sp) == 0
import numpy as np
a = np.array(list(sp), dtype=np.int64).reshape(-1, 2)
a = a[(a[:, 0] > 0) & (a[:, 1] > a[:, 0])]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # us
dur = en - st
print(f"{len(a)} workgroups: launch span {en.max():.1f} us; starts {st.min():.1f} .. {st.max():.1f} us (median {np.median(st):.1f}); ends {en.min():.1f} .. {en.max():.1f} (median {np.median(en):.1f}); "
      f"durations {dur.min():.1f} .. {dur.max():.1f} (median {np.median(dur):.1f})")
for x in range(8):
    m = np.arange(len(a)) % 8 == x
    print(f"  XCD {x}: start {st[m].mean():.1f}, duration {dur[m].mean():.1f} (max {dur[m].max():.1f}), end max {en[m].max():.1f}")
