"""conv1x1_big_kernel (256 x 256 tile, bd_conv_desc.route[0] mode 2) and the default dispatch on balanced grids (one or two tiles per CU) at growing K:
the slope is the cost of a 64-channel K block per CU, the intercept prologue + epilogue.   python scripts/exp/gemm_kslope.py"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_here)))
import torch
from basedet_amd import ops


def timeit(run, iters=30):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    modes = [int(a) for a in sys.argv[1:]] or [2, 3]
    for M, CO in ((65536, 256), (32768, 512), (8192, 2048), (16384, 2048)):
        tiles = (M // 256) * (CO // 256)
        print(f"-- M {M} x N {CO}: {tiles} tiles of 256 x 256")
        prev = {}
        for K in (256, 512, 1024, 2048, 4096):
            geo = ops.single(1, 1, M)
            d = ops.conv_desc(geo, geo, K, CO, 1, 1, 1, 0)
            x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(CO, 1, K, device="cuda") * 0.03).to(torch.bfloat16)
            y = torch.empty(M, CO, device="cuda", dtype=torch.bfloat16)
            row = []
            for m in modes:
                ops.set_route(dense1x1=m)
                us = timeit(lambda: ops.conv2d_fwd(d, x, w, None, y, flags=ops.EPI_RELU))
                name = ops.L().bd_conv_last_kernel().decode()
                dk = ""
                if m in prev:
                    pk, pus = prev[m]
                    dk = f" (+{(us - pus) / ((K - pk) / 64) / (tiles / 256):5.2f} us per K block and tile round)"
                prev[m] = (K, us)
                row.append(f"mode {m} {name:22s} {us:7.1f} us {2.0 * M * K * CO / us / 1e6:6.0f} TF/s {(2.0 * M * (K + CO) + 2.0 * K * CO) / us / 1e3:6.0f} GB/s{dk}")
            print(f"   K {K:5d}: " + " | ".join(row), flush=True)
            del x, w, y
    ops.set_route(dense1x1=1)


if __name__ == "__main__":
    main()
