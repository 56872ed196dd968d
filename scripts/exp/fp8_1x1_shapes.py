"""The dense 1x1 launches of RetinaNet-R101 at batch 32 on one-byte operands: conv1x1_fp8_kernel (bd_conv_desc.route[0] = 3) against the ring
kernel's one-byte form (5 = every legal launch) and the default rule (1).   python scripts/exp/fp8_1x1_shapes.py [iters=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basedet_amd import ops

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = int(os.environ.get("F8_N", 32))
# stage, H, W, bottleneck width C (block output 4 C)
STAGES = [("res3", 100, 168, 128), ("res4", 50, 84, 256), ("res5", 25, 42, 512)]


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS * 1e3


for stage, H, W, C in STAGES:
    M = N * H * W
    geo = ops.single(N, H, W)
    for name, mode, K, CO in (("conv1 fwd   4C->C  relu", 0, 4 * C, C), ("conv3 fwd   C->4C  add relu", 0, C, 4 * C),
                              ("conv3 dgrad 4C->C  gate", 1, 4 * C, C), ("conv1 dgrad C->4C  acc gate", 1, C, 4 * C)):
        if K % 128:
            continue
        d = ops.conv_desc(geo, geo, K, CO, 1, 1, 1, 0) if mode == 0 else ops.conv_desc(geo, geo, CO, K, 1, 1, 1, 0)
        xq = torch.randint(0, 120, (M * K,), dtype=torch.uint8, device="cuda")
        wq = torch.randint(0, 120, (CO, 1, K), dtype=torch.uint8, device="cuda")
        ws = torch.full((CO,), 2.0 ** -6, device="cuda")
        bias = torch.randn(CO, device="cuda")
        y = torch.empty((M, CO), dtype=torch.bfloat16, device="cuda")
        res = torch.randn(M, CO, device="cuda").to(torch.bfloat16)
        bits = torch.zeros((CO // 32, M), dtype=torch.int32, device="cuda")
        gate = torch.randint(-2 ** 31, 2 ** 31 - 1, (CO // 32, M), device="cuda", dtype=torch.int64).to(torch.int32)
        y8 = torch.empty((M, CO), dtype=torch.uint8, device="cuda")
        if mode == 0:
            add = res if "add" in name else None
            fn = lambda: ops.conv1x1_fp8(d, 0, xq, wq, ws, bias, y, add=add, bits=bits, y8=y8, q_scale=1.0,
                                         flags=ops.EPI_RELU | (ops.EPI_ADD_BEFORE if add is not None else 0))
            nbytes = M * K + K * CO + 2.0 * M * CO * (2 if add is not None else 1) + M * CO / 8 + M * CO
        else:
            acc = "acc" in name
            fn = lambda: ops.conv1x1_fp8(d, 1, xq, wq, ws, None, y, add=y if acc else None, maskbits=gate, y8=y8, q_scale=1.0,
                                         flags=ops.EPI_ADD_BEFORE if acc else 0)
            nbytes = M * K + K * CO + 2.0 * M * CO * (2 if acc else 1) + M * CO / 8 + M * CO
        row = []
        for route in (3, 5, None):
            ops.set_route(dense1x1=route)
            us = timeit(fn)
            row.append((us, ops.L().bd_conv_last_kernel().decode()))
        ops.set_route(dense1x1=None)
        print(f"{stage} {name:28s} M={M:7d} K={K:5d} CO={CO:5d}  dense {row[0][0]:7.1f} us ({nbytes / row[0][0] / 1e6:5.2f} TB/s)   ring {row[1][0]:7.1f} us "
              f"({nbytes / row[1][0] / 1e6:5.2f} TB/s)   default {row[2][0]:7.1f} us [{row[2][1]}]", flush=True)
