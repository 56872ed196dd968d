"""Workgroup timelines of conv1x1_dense_kernel launches (diagnostic build:
    BD_LIB_NAME=libbasedet_d1.so BD_EXTRA_FLAGS=-DBD_D1_STAMP python -m basedet_amd.build
run with BASEDET_HIP_LIB pointing at it).  Every workgroup stamps the 100 MHz clock at entry, behind the K loop and once its stores are
acknowledged, plus the CU it ran on.  Printed per launch: the kernel's span, the lifetime of a workgroup and its split, how many
workgroups a CU holds over time, and the bytes the chip moves per 5 us slice (ramp / plateau / tail)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from basedet_amd import ops, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from micro_1x1_step import L, make

ABLATES = [int(a) for a in os.environ.get("D1_ABLATES", "0").split(",")]     # BD_D1_ABLATE values (timing only; see P1::dbg)
WANT = sys.argv[1:] or ["res3 conv1 fwd", "res3 conv3 fwd", "res4 conv1 fwd", "res4 conv3 fwd", "res5 conv1 fwd", "res5 conv3 fwd", "res4 conv1 dgrad"]
lib = ctypes.CDLL(_lib.LIB_PATH)


def analyse(tag, st, n_tiles_n, nbytes, us_event):
    t0, t1, t2, hw = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64), st[:, 2].astype(np.int64), st[:, 3]
    base = t0.min()
    t0, t1, t2 = (t0 - base) * 10, (t1 - base) * 10, (t2 - base) * 10          # ns
    span = t2.max()
    hwid = (hw & 0xffffffff).astype(np.int64)
    xcc = (hw >> 32).astype(np.int64) & 0xf
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cus = np.unique(key)
    life = t2 - t0
    print(f"== {tag}: {len(st)} workgroups on {len(cus)} CUs; span {span / 1e3:.1f} us (events: {us_event:.1f} us per launch) -> {nbytes / span:.0f} GB/s over the span")
    print(f"   workgroup lifetime: mean {life.mean() / 1e3:.2f} us (p10 {np.percentile(life, 10) / 1e3:.2f}, p90 {np.percentile(life, 90) / 1e3:.2f});"
          f" K loop {np.mean(t1 - t0) / 1e3:.2f} us, epilogue {np.mean(t2 - t1) / 1e3:.2f} us")
    # residency: sum of lifetimes / (CUs x span) = average workgroups per CU
    print(f"   average residency {life.sum() / (len(cus) * span):.2f} workgroups per CU; per tile and CU {span * len(cus) / len(st) / 1e3:.2f} us")
    first = np.sort(t0)[: len(cus)]
    print(f"   launch ramp: workgroup #{len(cus)} started at {first[-1] / 1e3:.2f} us; the last workgroup started at {t0.max() / 1e3:.2f} us, "
          f"ended {span / 1e3:.2f} us; the first ended at {t2.min() / 1e3:.2f} us")
    # chip throughput per slice: bytes of a workgroup spread evenly over its lifetime
    nb = 12
    edges = np.linspace(0, span, nb + 1)
    per_wg = nbytes / len(st)
    row, occ = [], []
    for a, b in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(t2, b) - np.maximum(t0, a), 0, None)
        row.append((ov / np.maximum(life, 1)).sum() * per_wg / (b - a))
        occ.append(ov.sum() / ((b - a) * len(cus)))
    print("   GB/s per slice:      " + " ".join(f"{v:6.0f}" for v in row))
    print("   workgroups per CU:   " + " ".join(f"{v:6.2f}" for v in occ))
    # workgroups per XCD (is the remap balanced) and lifetimes by start order (first round vs steady state)
    order = np.argsort(t0)
    q = len(st) // 4
    print("   lifetime by start quartile (us): " + " ".join(f"{life[order[i * q:(i + 1) * q]].mean() / 1e3:.2f}" for i in range(4)))
    print("   K loop by start quartile (us):   " + " ".join(f"{(t1 - t0)[order[i * q:(i + 1) * q]].mean() / 1e3:.2f}" for i in range(4)))
    per_xcc = [int((xcc == k).sum()) for k in range(8)]
    print(f"   workgroups per XCD: {per_xcc}")


for tag, H, W, Cin, Cout, mode, add, mask, cnt in L:
    if not any(tag.startswith(w) for w in WANT):
        continue
    run, nbytes, _ = make(H, W, Cin, Cout, mode, add, mask, 1)
    for ab in ABLATES:
        os.environ["BD_D1_ABLATE"] = str(ab)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        M = 16 * H * W
        cd = Cout if mode == "fwd" else Cin
        grid = ((M + 127) // 128) * ((cd + 127) // 128)
        n = min(grid, 32768)
        out = (ctypes.c_ulonglong * (4 * n))()
        assert lib.bd_debug_d1_stamp(out, n) == 0
        st = np.frombuffer(out, dtype=np.uint64).reshape(n, 4).copy()
        analyse(tag + (f"  [ablate {ab}: " + ", ".join(n_ for b_, n_ in ((1, "no stores"), (2, "no epilogue loads"), (4, "pixel tile 0 only"), (8, "no K loop")) if ab & b_) + "]" if ab else ""),
                st, (cd + 127) // 128, nbytes, us)
    del run
    torch.cuda.empty_cache()
