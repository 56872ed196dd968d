"""The dense 1x1 launches of one RetinaNet-R50 step (800x1344, batch 16) with their epilogue operands, timed per bd_conv_desc.route[0]
mode (0 = the generic kernel, 1 = default choice, 2 = the 256^2 LDS-DMA tile wherever legal, 3 = the 128^2 tile only, 5 = conv1x1_ring_kernel wherever legal): algorithmic GB/s
per launch class.   python scripts/micro_1x1_step.py [modes...]"""
import os
import sys
_here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_here))
import torch
from basedet_amd import ops

N = 16
# (tag, H, W, Cin, Cout, mode, add, mask, launches per step)
L = [
    ("res2 conv1 fwd 256->64 relu", 200, 336, 256, 64, "fwd", 0, 0, 2),
    ("res2 conv3 fwd 64->256 +add relu", 200, 336, 64, 256, "fwd", 1, 0, 3),
    ("res3 conv1 fwd 512->128", 100, 168, 512, 128, "fwd", 0, 0, 3),
    ("res3 conv3 fwd 128->512 +add", 100, 168, 128, 512, "fwd", 1, 0, 4),
    ("res4 conv1 fwd 1024->256", 50, 84, 1024, 256, "fwd", 0, 0, 5),
    ("res4 conv3 fwd 256->1024 +add", 50, 84, 256, 1024, "fwd", 1, 0, 6),
    ("res5 conv1 fwd 2048->512", 25, 42, 2048, 512, "fwd", 0, 0, 2),
    ("res5 conv3 fwd 512->2048 +add", 25, 42, 512, 2048, "fwd", 1, 0, 3),
    ("res3 conv3 dgrad 512->128 mask", 100, 168, 128, 512, "dgrad", 0, 1, 4),
    ("res3 conv1 dgrad 128->512 add+mask", 100, 168, 512, 128, "dgrad", 1, 1, 3),
    ("res4 conv3 dgrad 1024->256 mask", 50, 84, 256, 1024, "dgrad", 0, 1, 6),
    ("res4 conv1 dgrad 256->1024 add+mask", 50, 84, 1024, 256, "dgrad", 1, 1, 5),
    ("res5 conv3 dgrad 2048->512 mask", 25, 42, 512, 2048, "dgrad", 0, 1, 3),
    ("res5 conv1 dgrad 512->2048 add+mask", 25, 42, 2048, 512, "dgrad", 1, 1, 2),
    ("lateral3 fwd 512->256", 100, 168, 512, 256, "fwd", 0, 0, 1),
    ("lateral4 dgrad 256->1024", 50, 84, 1024, 256, "dgrad", 0, 0, 1),
]


def make(H, W, Cin, Cout, mode, add, mask, bits):
    geo = ops.single(N, H, W)
    d = ops.conv_desc(geo, geo, Cin, Cout, 1, 1, 1, 0)
    M = geo.pixels
    cs, cd = (Cin, Cout) if mode == "fwd" else (Cout, Cin)
    src = torch.randn(M, cs, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cd, 1, cs, device="cuda") * 0.03).to(torch.bfloat16)
    dst = torch.empty(M, cd, device="cuda", dtype=torch.bfloat16)
    addt = torch.randn(M, cd, device="cuda").to(torch.bfloat16) if add else None
    maskt = torch.randn(M, cd, device="cuda").to(torch.bfloat16) if mask else None
    mb = torch.randint(-2 ** 31, 2 ** 31 - 1, (cd // 32, M), device="cuda", dtype=torch.int32) if (mask and bits) else None
    yb = torch.empty((cd // 32, M), device="cuda", dtype=torch.int32) if (mode == "fwd" and add and bits) else None
    nbytes = 2.0 * M * (cs + cd) + 2.0 * cs * cd + (2.0 * M * cd if add else 0) + ((M * cd / 8.0 if mb is not None else 2.0 * M * cd) if mask else 0) \
        + (M * cd / 8.0 if yb is not None else 0)

    def run():
        if mode == "fwd":
            ops.conv2d_fwd(d, src, w, None, dst, add=addt, flags=ops.EPI_RELU | (ops.EPI_ADD_BEFORE if add else 0), bits=yb)
        else:
            fl = (ops.EPI_ADD_BEFORE if add else 0) | (ops.EPI_MASK if mask else 0)
            ops.conv2d_dgrad(d, src, w, dst, add=addt, mask=None if mb is not None else maskt, maskbits=mb, flags=fl)
    return run, nbytes, 2.0 * M * Cin * Cout


def timeit(run, iters=20):
    for _ in range(3):
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
    depths = [int(a) for a in sys.argv[1:]] or [3, 2]
    if os.environ.get("BD_KNOB"):          # a route[1] bit mask for the run
        ops.set_route(patch3x3=int(os.environ["BD_KNOB"]))
    tot = {(dp, b): 0.0 for dp in depths for b in (0, 1)}
    print(f"{'launch':40s} " + " ".join(f"d{dp}{'b' if b else ' '}:us/GB/s" .rjust(16) for dp in depths for b in ((0, 1) if dp else (0,))))
    for tag, H, W, Cin, Cout, mode, add, mask, cnt in L:
        row = []
        for dp in depths:
            for b in ((0, 1) if dp else (0,)):
                ops.set_route(dense1x1=dp)
                run, nb, fl = make(H, W, Cin, Cout, mode, add, mask, b)
                us = timeit(run)
                tot[(dp, b)] += us * cnt
                row.append(f"{us:7.1f}/{nb / us / 1e3:6.0f}")
                del run
                torch.cuda.empty_cache()
        print(f"{tag:40s} " + " ".join(r.rjust(16) for r in row), flush=True)
    print("sum over the step's launches (ms): " + ", ".join(f"d{dp}{'b' if b else ''}={v / 1e3:.3f}" for (dp, b), v in tot.items() if dp or not b))
    ops.set_route(dense1x1=1)


if __name__ == "__main__":
    main()
