"""Dense 1x1 launches as plain GEMMs on well- and badly-quantised shapes: which bd_conv_desc.route[0] mode runs them how fast.
   python scripts/exp/gemm_probe.py [modes...]   (default 1 2: the default dispatch, conv1x1_big_kernel everywhere; modes 7 / 8 existed while the K-sliced kernel did:
profiles/r06_dense1x1_sk.txt)"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_here)))
import torch
from basedet_amd import ops

# (tag, M pixels, K, Cout)
SHAPES = [
    ("256 tiles  K1024 N256", 65536, 1024, 256),
    ("512 tiles  K1024 N256", 131072, 1024, 256),
    ("res4 conv1 67200 K1024 N256", 67200, 1024, 256),
    ("256 tiles  K2048 N512", 32768, 2048, 512),
    ("res5 conv1 16800 K2048 N512", 16800, 2048, 512),
    ("res5 conv3 16800 K512 N2048", 16800, 512, 2048),
    ("512 tiles  K512 N2048", 16384, 512, 2048),
    ("lateral5 16800 K2048 N256", 16800, 2048, 256),
    ("res5.0 conv1 67200 K1024 N512", 67200, 1024, 512),
]


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
    modes = [int(a) for a in sys.argv[1:]] or [1, 2]
    print(f"{'shape':34s} " + " ".join(f"mode{m}: us / TF/s / kernel".rjust(44) for m in modes))
    for tag, M, K, CO in SHAPES:
        geo = ops.single(1, 1, M)
        d = ops.conv_desc(geo, geo, K, CO, 1, 1, 1, 0)
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(CO, 1, K, device="cuda") * 0.03).to(torch.bfloat16)
        y = torch.empty(M, CO, device="cuda", dtype=torch.bfloat16)
        row = []
        ref = None
        for m in modes:
            ops.set_route(dense1x1=m)
            run = lambda: ops.conv2d_fwd(d, x, w, None, y, flags=ops.EPI_RELU)
            us = timeit(run)
            name = ops.L().bd_conv_last_kernel().decode()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
                same = ""
            else:
                same = " same" if torch.equal(ref, y) else f" DIFF {(ref.float() - y.float()).abs().max().item():.3g}"
            row.append(f"{us:7.1f} / {2.0 * M * K * CO / us / 1e6:6.0f} / {name}{same}")
        print(f"{tag:34s} " + " ".join(r.rjust(44) for r in row), flush=True)
        del x, w, y
    ops.set_route(dense1x1=1)


if __name__ == "__main__":
    main()
