#!/usr/bin/env python3
"""Headline benchmark: images/sec, training RetinaNet-R50-FPN on synthetic 1333x800 (padded 800x1344) images.

One "step" = one full training step (pre-process, forward, target assignment, losses, backward, gradient all-reduce, SGD + weight
repack) over one DummyLoader-shaped batch (basedet/utils/dummy.py:8-63).  The headline `value` times K steps between two device
syncs (+ barrier for N > 1) with the batch ALREADY RESIDENT IN HBM (the fp32 NCHW -> bf16 NHWC conversion is inside the step).
`reference_protocol` times the same step the way the reference's own harness does (basedet/tools/benchmark.py:125-140): the
float64 host batch goes to the device inside the timed step and the device is synchronised before and after EVERY step; it is a
second, PCIe-inclusive figure -- never `value`.

One process per GPU.  `python bench.py --gpus N` starts the N rank processes itself (fresh children, before the parent touches the
GPU; like the reference's harness, tools/benchmark.py:269); under `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` the launcher's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are used as they are.  Collectives = bd_comm_* (RCCL over xGMI).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the kernel with the largest share of the step (per-kernel entries: roofline_others).  Each kernel is priced against
                  the roof its launches sit under: "mfma" (algorithmic FLOPs / time vs 2.5 PFLOP/s) when their arithmetic intensity
                  exceeds the machine balance, else "hbm" (algorithmic bytes -- every operand once -- / time vs 8 TB/s); durations
                  are HIP events on the launch stream during the timed steps, `traffic` the PMC-measured HBM bytes per launch:
                  at N = 1 measured IN the run (after the timed region rank 0 starts `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
                  child processes around three steps of the same workload: measure_traffic; --no-pmc skips them), else read from
                  profiles/r03_pmc_traffic.json (refused -- null -- when that was collected on a different build of the kernels)
  cpu_baseline -- the CPU oracle's (oracle/model.py, torch-CPU fp32) training step on the host cores, bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# HIP binds a stream to one of GPU_MAX_HW_QUEUES (default 4) hardware queues at its first use, sharing queues beyond that.  With
# RCCL's streams up the weight-gradient side stream ended up on the MAIN stream's queue: the two ran back to back (rocprofv3 kernel
# trace, Queue_Id column; scripts/queue_map.py).  Eight queues keep them apart.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense")
PEAK_FP8_TFLOPS = 5000.0      # dense fp8 (same table)
PEAK_HBM_GBS = 8000.0         # HBM3E, ~8 TB/s (same guide)
TRAIN_GFLOP_PER_IMG = {       # BASELINE.md section 3 / SURVEY.md section 8(d): convs only, 1 MAC = 2 FLOP
    "retinanet_r50_800x1344": 1435.6,
    "retinanet_r18_512x512": 277.2,
    "fcos_r50_800x1344": 1227.8,
    "retinanet_r101_800x1344": 1912.8,
    "atss_r50_800x1344": 1227.8,          # the FCOS network; only the target assignment differs
    "freeanchor_r50_800x1344": 1435.6,    # the RetinaNet network; bag losses instead of matcher + focal / L1
    "ota_r50_800x1344": 1227.8,           # the FCOS network; prediction-aware dynamic top-k assignment
    # Faster R-CNN R50-FPN (P2-P6): fwd 208.9 GMAC/img (backbone 87.6, FPN 60.9, RPN 53.1, box head 512 RoIs x 14.3 MMAC = 7.3);
    # stem + layer1 frozen, lateral2 needs no dgrad
    "faster_rcnn_r50_800x1344": 1177.2,
}

torch = None        # imported by the worker only: the launching parent never touches torch or the GPU
np = None


class KernelTimer:
    """HIP-event timing + algorithmic FLOP / byte accounting of the conv launches (fwd / dgrad / wgrad) and of the HBM-bound
    elementwise kernels SURVEY section 8(d) names (focal loss, SGD, stem, max-pool, pad + normalise)."""

    def __init__(self, ops):
        self.ops = ops
        self.records = {}
        self.enabled = False
        self.meta = {}

        def flops(d):
            m = sum(d.Ho[i] * d.Wo[i] for i in range(d.nseg)) * d.N
            return 2.0 * m * d.Cin * d.Cout * d.R * d.S

        def last_kernel():
            """The kernel the launch that just returned was dispatched to, as named by the launch site itself (bd_conv_last_kernel,
            include/basedet_hip.h): rounds 1-4 mirrored the C++ dispatch in Python here, and a dispatch change not mirrored would have
            mislabelled a row silently (tests/test_conv_gpu.py pins the names for the bench's descriptors)."""
            return ops.L().bd_conv_last_kernel().decode()

        def wrap(fn, kind):
            import inspect
            names = list(inspect.signature(fn).parameters)

            def inner(d, *a, **k):
                if not self.enabled:
                    return fn(d, *a, **k)
                k = dict(zip(names[1:], a), **k)         # all operands by name (add= / mask= may come positionally)
                a = ()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(d, *a, **k)
                e.record()
                kern = last_kernel()
                self.records.setdefault(kern, []).append((s, e, flops(d)))
                # algorithmic HBM bytes: every operand once -- both activations, the weights, and the epilogue's add / mask
                # operands (residual, ReLU mask), which have the shape of the result
                mi = sum(d.Hi[i] * d.Wi[i] for i in range(d.nseg)) * d.N
                mo = sum(d.Ho[i] * d.Wo[i] for i in range(d.nseg)) * d.N
                if d.R * d.S == 1 and d.stride == 2 and (fn.__name__ != "conv2d_dgrad" or (int(k.get("flags") or 0) & ops.EPI_SPARSE)):
                    mi = mo          # a 1x1 / stride-2 launch touches the quarter of the big grid's pixels it reaches (forward and weight-gradient reads, sparse dgrad writes)
                nbytes = 2.0 * (mi * d.Cin + mo * d.Cout) + (4.0 if kind == "wgrad" else 2.0) * d.Cin * d.Cout * d.R * d.S
                if kind == "igemm":
                    res = 2.0 * (mi * d.Cin if fn.__name__ == "conv2d_dgrad" else mo * d.Cout)
                    nbytes += res * ((k.get("add") is not None) + (k.get("mask") is not None))
                    if k.get("maskbits") is not None:
                        nbytes += res / 16.0               # the bit-packed ReLU gate: 1 bit per element
                    if k.get("bits") is not None:
                        nbytes += res / 16.0
                self.meta.setdefault(kern, []).append((fn.__name__, d.Cin, d.Cout, d.R, d.stride, d.nseg, d.Ho[0], d.Wo[0], nbytes))
                return r
            return inner

        def wrap_fp8(fn):
            def inner(d, xq, wq, wscale, bias, y, add=None, flags=0, y8=None, q_scale=1.0):
                if not self.enabled:
                    return fn(d, xq, wq, wscale, bias, y, add=add, flags=flags, y8=y8, q_scale=q_scale)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(d, xq, wq, wscale, bias, y, add=add, flags=flags, y8=y8, q_scale=q_scale)
                e.record()
                mi = sum(d.Hi[i] * d.Wi[i] for i in range(d.nseg)) * d.N
                mo = sum(d.Ho[i] * d.Wo[i] for i in range(d.nseg)) * d.N
                nbytes = 1.0 * mi * d.Cin + 2.0 * mo * d.Cout * (2 if add is not None else 1) + 1.0 * d.Cin * d.Cout * d.R * d.S \
                    + (1.0 * mo * d.Cout if y8 is not None else 0.0)
                kern8 = last_kernel()
                self.records.setdefault(kern8, []).append((s, e, flops(d)))
                self.meta.setdefault(kern8, []).append(("conv2d_fwd_fp8", d.Cin, d.Cout, d.R, d.stride, d.nseg, d.Ho[0], d.Wo[0], nbytes))
                return r
            return inner

        def wrap_fp8_dgrad(fn):
            def inner(d, g8, wq_t, wscale_t, dx, add=None, mask=None, flags=0, dx8=None, q_scale=1.0):
                if not self.enabled:
                    return fn(d, g8, wq_t, wscale_t, dx, add=add, mask=mask, flags=flags, dx8=dx8, q_scale=q_scale)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(d, g8, wq_t, wscale_t, dx, add=add, mask=mask, flags=flags, dx8=dx8, q_scale=q_scale)
                e.record()
                mi = sum(d.Hi[i] * d.Wi[i] for i in range(d.nseg)) * d.N
                mo = sum(d.Ho[i] * d.Wo[i] for i in range(d.nseg)) * d.N
                nbytes = 1.0 * mo * d.Cout + 2.0 * mi * d.Cin * (1 + (add is not None) + (mask is not None)) + 1.0 * d.Cin * d.Cout * 9 \
                    + (1.0 * mi * d.Cin if dx8 is not None else 0.0)
                kern8 = last_kernel()
                self.records.setdefault(kern8, []).append((s, e, flops(d)))
                self.meta.setdefault(kern8, []).append(("conv2d_dgrad_fp8", d.Cin, d.Cout, d.R, d.stride, d.nseg, d.Ho[0], d.Wo[0], nbytes))
                return r
            return inner

        def wrap_fp8_1x1(fn):
            def inner(d, mode, xq, wq, wscale, bias, y, add=None, mask=None, maskbits=None, bits=None, y8=None, q_scale=1.0, flags=0):
                kw = dict(add=add, mask=mask, maskbits=maskbits, bits=bits, y8=y8, q_scale=q_scale, flags=flags)
                if not self.enabled:
                    return fn(d, mode, xq, wq, wscale, bias, y, **kw)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(d, mode, xq, wq, wscale, bias, y, **kw)
                e.record()
                m = d.Hi[0] * d.Wi[0] * d.N
                K, CO = (d.Cin, d.Cout) if mode == 0 else (d.Cout, d.Cin)
                # one-byte operand + weights, bf16 output (+ residual, + bf16 or bit-packed gate), optional twin / gate bits out
                nbytes = 1.0 * m * K + 1.0 * K * CO + 2.0 * m * CO * (1 + (add is not None) + (mask is not None and maskbits is None)) \
                    + (m * CO / 8.0 if maskbits is not None else 0.0) + (m * CO / 8.0 if bits is not None else 0.0) \
                    + (1.0 * m * CO if y8 is not None else 0.0)
                kern = ops.L().bd_conv_last_kernel().decode() or "conv1x1_fp8_kernel"      # conv1x1_fp8_kernel or conv1x1_ring_fp8_kernel (round 6)
                self.records.setdefault(kern, []).append((s, e, flops(d)))
                self.meta.setdefault(kern, []).append(("conv1x1_fp8 " + ("fwd" if mode == 0 else "dgrad"), d.Cin, d.Cout, 1, 1, 1, d.Ho[0], d.Wo[0], nbytes))
                return r
            return inner

        ops.conv1x1_fp8 = wrap_fp8_1x1(ops.conv1x1_fp8)
        ops.conv2d_dgrad_fp8 = wrap_fp8_dgrad(ops.conv2d_dgrad_fp8)
        ops.conv2d_fwd_fp8 = wrap_fp8(ops.conv2d_fwd_fp8)
        ops.conv2d_fwd = wrap(ops.conv2d_fwd, "igemm")
        ops.conv2d_dgrad = wrap(ops.conv2d_dgrad, "igemm")
        ops.conv2d_wgrad = wrap(ops.conv2d_wgrad, "wgrad")
        ops.conv2d_wgrad_bias = wrap(ops.conv2d_wgrad_bias, "wgrad")

        # ---- HBM-bound elementwise kernels: algorithmic bytes per launch from the call's own arguments
        def wrap_stream(name, kern, nbytes_of):
            fn = getattr(ops, name, None)
            if fn is None:
                return
            import inspect
            names = list(inspect.signature(fn).parameters)

            def inner(*a, **k):
                if not self.enabled:
                    return fn(*a, **k)
                kw = dict(zip(names, a), **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(*a, **k)
                e.record()
                self.records.setdefault(kern, []).append((s, e, 0.0))
                self.meta.setdefault(kern, []).append((name, 0, 0, 0, 0, 0, 0, 0, float(nbytes_of(kw))))
                return r
            setattr(ops, name, inner)

        # focal: 2 B logit read + 2 B gradient write per logit, 4 B label per row
        wrap_stream("focal_loss_fwd_bwd", "focal_g2_kernel", lambda k: k["rows"] * k["K"] * 4.0 + k["rows"] * 4.0)
        # SGD: read w, v, g; write w, v (fp32)
        wrap_stream("sgd_momentum_step", "sgd_kernel", lambda k: k["w"].numel() * 20.0)
        # stem: bf16 [N][H+6][W+8][4] in, bf16 [N][H/2][W/2][64] out
        wrap_stream("stem_conv7x7_fwd", "stem_conv_kernel",
                    lambda k: k["N"] * ((k["H"] + 6) * (k["W"] + 8) * 8.0 + (k["H"] // 2) * (k["W"] // 2) * 128.0))
        wrap_stream("stem_pool_fwd", "stem_pool_kernel",
                    lambda k: k["N"] * ((k["H"] + 6) * (k["W"] + 8) * 8.0 + ((k["H"] // 2 - 1) // 2 + 1) * ((k["W"] // 2 - 1) // 2 + 1) * 128.0))
        wrap_stream("maxpool3x3s2_fwd", "maxpool3x3s2_kernel",
                    lambda k: k["N"] * k["Cn"] * 2.0 * (k["H"] * k["W"] + ((k["H"] - 1) // 2 + 1) * ((k["W"] - 1) // 2 + 1)))
        # pad + normalise: fp32 NCHW in (3 channels), bf16 [N][Hp+6][Wp+8][4] out
        # e4m3 cast of a convolution's input: 2 B read + 1 B written per element
        wrap_stream("quantize_fp8", "quantize_fp8_kernel", lambda k: k["x"].numel() * 3.0)
        wrap_stream("quantize_bf8", "quantize_bf8_kernel", lambda k: k["x"].numel() * 3.0)
        # fused frozen bottleneck: block input once + block output once (+ the weights)
        wrap_stream("bottleneck_fwd", "bottleneck_fused_kernel",
                    lambda k: k["N"] * k["H"] * k["W"] * (k["cin"] + k["cout"]) * 2.0
                    + 2.0 * (k["cin"] * k["cmid"] + 9 * k["cmid"] * k["cmid"] + k["cmid"] * k["cout"] + (k["cin"] * k["cout"] if k["wd"] is not None else 0)))
        # GroupNorm + ReLU of the FCOS towers (bd_groupnorm_fwd: statistics + apply; _bwd: sums + apply; every operand once: y, z / dz, y, dy)
        wrap_stream("groupnorm_fwd", "gn_stats + gn_apply (bd_groupnorm_fwd)", lambda k: k["y"].numel() * 2.0 * 2)
        wrap_stream("groupnorm_bwd", "gn_bwd_partial + gn_bwd_apply (bd_groupnorm_bwd)", lambda k: k["y"].numel() * 2.0 * 3)
        # Faster R-CNN (round 5).  RPN prediction layer on its own kernels: x once (+ dx once) + the 16-channel side; RoIAlign backward as a
        # tiled sum: the pooled gradients once + the touched part of the gradient pyramid (read + written; counted as the pooled bytes again)
        wrap_stream("conv1x1_thin_fwd", "conv1x1_thin_fwd_kernel", lambda k: k["M"] * (k["Cin"] + k["Cout"]) * 2.0)
        wrap_stream("conv1x1_thin_bwd", "conv1x1_thin_bwd_kernel", lambda k: k["M"] * (2 * k["Cin"] + k["Cout"]) * 2.0)
        wrap_stream("roi_align_bwd_bf16", "roi_align_bwd_tile_kernel", lambda k: k["gout"].numel() * 2.0 * 2)
        wrap_stream("roi_align_fwd", "roi_align_fwd_kernel", lambda k: k["out"].numel() * 2.0 * 2)
        wrap_stream("pad_normalize", "pad_normalize_kernel",
                    lambda k: k["x"].numel() * 4.0 + k["x"].shape[0] * (k["Hp"] + 6) * (k["Wp"] + 8) * 8.0)

    def dump(self, steps):
        agg = {}
        for kind, rec in self.records.items():
            for (s, e, f), m in zip(rec, self.meta.get(kind, [])):
                k = (kind,) + m[:8]
                a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += s.elapsed_time(e); a[2] += f; a[3] += m[8]
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        print("# per-shape kernel time per step (ms), TFLOP/s", file=sys.stderr)
        for k, (n, ms, fl, nb) in rows:
            print(f"# {k[0]:22s} {k[1]:14s} Cin={k[2]:5d} Cout={k[3]:5d} R={k[4]} s={k[5]} nseg={k[6]} HxW={k[7]}x{k[8]} "
                  f"launches/step={n // steps:3d} ms/step={ms / steps:7.3f} TF/s={fl / (ms * 1e-3) / 1e12:7.1f} "
                  f"minGB/s={nb / (ms * 1e-3) / 1e9:7.0f}", file=sys.stderr)

    def summary(self, kind):
        rec = self.records[kind]
        if not rec:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _ in rec)
        fl = sum(f for _, _, f in rec)
        nb = sum(m[8] for m in self.meta[kind])
        return dict(launches=len(rec), ms=ms, flops=fl, bytes=nb)

    def by_launch_class(self, kind, ridge):
        """The kernel's launches split by EACH LAUNCH's own arithmetic intensity against the machine balance `ridge` (FLOP per byte): a
        kernel whose average launch is HBM-bound can hold launches that sit above the ridge (res5's conv1: 400 FLOP/B) -- they are graded
        against the MFMA roof, the others against HBM.  {"mfma": {...}, "hbm": {...}} with launches, ms, flops, bytes per class."""
        out = {}
        for (s, e, f), m in zip(self.records[kind], self.meta[kind]):
            cls = "mfma" if (m[8] > 0 and f / m[8] >= ridge) else "hbm"
            a = out.setdefault(cls, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1; a["ms"] += s.elapsed_time(e); a["flops"] += f; a["bytes"] += m[8]
        return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle_step(orc, b, lr, wd, min_iters, seconds):
    """One warm-up iteration, then >= min_iters timed iterations (more while the budget lasts, at most 6)."""
    state, t_all = {}, []
    t_start = time.time()
    it = 0
    while True:
        t0 = time.time()
        losses, _ = orc.retinanet_losses(b)
        g = orc.grads(losses["total_loss"])
        orc.sgd_step(g, state, lr, 0.9, wd)
        dt = time.time() - t0
        it += 1
        if it > 1:
            t_all.append(dt)
        if len(t_all) >= min_iters and (time.time() - t_start > seconds or len(t_all) >= 6):
            break
    return t_all


def cpu_baseline(cfg, params, seconds=25.0, batch=2, size=(800, 1344)):
    """Oracle training step (fwd + bwd + SGD) on the host cores: a bounded sample of the headline workload (batch 2 at full
    resolution, >= 3 timed iterations -- SURVEY 8d) plus BASELINE config C1 (RetinaNet-R18, 2 x 512x512) in full."""
    from basedet_amd.configs import retinanet_r18_config
    from basedet_amd.models import params as P
    from basedet_amd.utils import DummyLoader
    from oracle.model import Oracle
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    b = next(DummyLoader(batch, size, seed=0))
    b["data"] = b["data"].astype(np.float32)
    t_all = _time_oracle_step(orc, b, cfg.SOLVER.BASIC_LR * batch, cfg.SOLVER.WEIGHT_DECAY, 3, seconds)
    mean = float(np.mean(t_all))
    out = dict(value=round(batch / mean, 4), unit="images/sec", cores=torch.get_num_threads(), kind="port", cpu=_cpu_model(),
               sample=f"oracle/model.py torch-CPU fp32 RetinaNet-R50 step, batch {batch} x {size[0]}x{size[1]}, "
                      f"{len(t_all)} timed iterations after 1 warm-up, {mean:.2f} s/iter (min {min(t_all):.2f}, max {max(t_all):.2f})")
    # C1: the reference's own CPU-runnable plumbing configuration
    c1 = retinanet_r18_config()
    p1 = P.init_retinanet_params(c1, seed=0)
    o1 = Oracle(p1, P.oracle_arch(c1), trainable=P.trainable_names(p1, c1.MODEL.BACKBONE.FREEZE_AT))
    b1 = next(DummyLoader(2, (512, 512), seed=0))
    b1["data"] = b1["data"].astype(np.float32)
    t1 = _time_oracle_step(o1, b1, c1.SOLVER.BASIC_LR * 2, c1.SOLVER.WEIGHT_DECAY, 3, 5.0)
    out["c1_retinanet_r18_2x512x512"] = dict(value=round(2 / float(np.mean(t1)), 3), unit="images/sec",
                                             sample=f"{len(t1)} timed iterations after 1 warm-up, {float(np.mean(t1)):.3f} s/iter")
    return out


def _stdout_to_stderr():
    """RCCL prints a version banner on the C stdout when the first communicator comes up (and it surfaces whenever that buffer is
    flushed -- after the JSON line, at exit): everything but the result line goes to stderr."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def _print_result(saved_fd, line):
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)         # C stdio buffers (the banner) leave through the redirected descriptor
    except OSError:
        pass
    os.dup2(saved_fd, 1)
    print(line, flush=True)
    os.dup2(2, 1)
    os.close(saved_fd)


def measure_traffic(args, timeout_s=120):
    """roofline.traffic measured IN this run: after the timed region rank 0 starts two child processes, `rocprofv3 --pmc FETCH_SIZE`
    and `rocprofv3 --pmc WRITE_SIZE` (separate passes, counters only: MI355X_MICROARCH.md, HBM section) around three steps of the same
    workload, and converts the per-kernel means to HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE
    counts 128-byte requests at 64 B).  Returns ({kernel: {"hbm_bytes_per_launch": n}}, note); ({}, reason) when it cannot run (no
    rocprofv3, already under a profiler, a pass failed or timed out) -- the caller then falls back to the committed file."""
    import shutil
    import tempfile
    if any(k.startswith(("ROCP_", "ROCPROFILER_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {}, "bench.py itself runs under a profiler: the in-run PMC passes were skipped"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found: traffic not measured"
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import pmc_traffic
    tmp = tempfile.mkdtemp(prefix="bd_pmc_", dir="/tmp")
    tail = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-pmc",
            "--ref-protocol-steps", "0", "--workload", args.workload, "--batch", str(args.batch)] + (["--fp8"] if args.fp8 else [])
    for kv in args.model_opt:
        tail += ["--model-opt", kv]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = {}
    try:
        clocks = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            # GRBM_GUI_ACTIVE rides with WRITE_SIZE (2 of the 4 TCC slots; the GRBM block has its own two): the clock the chip held in each
            # kernel of that pass = counter / 8 XCDs / dispatch time (MI355X_MICROARCH.md, DVFS give-back)
            names = [counter, "GRBM_GUI_ACTIVE"] if counter == "WRITE_SIZE" else [counter]
            # own process group: on a timeout the profiler AND the profiled child (which holds the GPU) are ended together, by that exact group id
            proc = subprocess.Popen([exe, "--pmc"] + names + ["--output-format", "csv", "-d", d, "--"] + tail, cwd="/tmp", env=env,
                                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            if rc != 0:
                return {}, f"rocprofv3 --pmc {counter} exited with {rc}: traffic not measured"
            res[counter] = pmc_traffic.load(d, counter)
            if counter == "WRITE_SIZE":
                try:
                    clocks = pmc_traffic.load_clock(d)
                except Exception:                                   # noqa: BLE001 (the clock is an extra: never lose the traffic over it)
                    clocks = {}
    except subprocess.TimeoutExpired:
        return {}, f"a rocprofv3 --pmc pass exceeded {timeout_s} s: traffic not measured"
    except Exception as e:                                          # noqa: BLE001 (a missing csv, a parse error: report, do not fail the bench)
        return {}, f"in-run PMC passes failed ({type(e).__name__}: {e}): traffic not measured"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (f, fc), (w, wc) = res["FETCH_SIZE"], res["WRITE_SIZE"]
    out = {k: {"hbm_bytes_per_launch": int((2 * f[k] / fc[k] + w.get(k, 0.0) / max(1, wc.get(k, 0))) * 1024)} for k in f}
    for k, (mhz, us) in clocks.items():
        if k in out:
            out[k]["clock_mhz"], out[k]["pmc_dispatch_us"] = round(mhz, 0), round(us, 1)
    return out, ("traffic measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two child processes after the timed region, "
                 "3 steps each, mean per launch), HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv=None, script=None, exit=True, poll_s=0.2):
    """The parent of a `python bench.py --gpus N` run: N fresh rank processes (one per GPU), started before this process has
    imported torch or made any HIP call; relays rank 0's result line; non-zero exit if any rank fails.  The children are ended by
    their exact PIDs if one of them dies (the survivors would wait in a collective forever).
    `argv` / `script` / `exit=False` (returns (rc, pids) instead of leaving): tests/test_dist_cpu.py drives it with a stand-in script."""
    port = _free_port()
    procs = []
    argv = sys.argv[1:] if argv is None else list(argv)
    script = os.path.abspath(__file__) if script is None else script
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # as torch.distributed.run does: N ranks with a full-size OpenMP pool each would oversubscribe the host's cores
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // n))))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    line = None
    import threading

    def _read():
        nonlocal line
        for ln in procs[0].stdout:
            if ln.strip().startswith("{"):
                line = ln.strip()

    t = threading.Thread(target=_read, daemon=True)
    t.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is not None:
                alive.discard(r)
                if code != 0:
                    rc = rc or code
                    for o in alive:          # one rank failed: the others would wait in a collective forever
                        procs[o].kill()
        time.sleep(poll_s)
    for p in procs:
        p.wait()
    t.join(5)
    if line:
        print(line, flush=True)
    if rc == 0 and not line:
        rc = 1
    if not exit:
        return rc, [p.pid for p in procs]
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # the reference harness: 100 iterations ...
    ap.add_argument("--warmup", type=int, default=20)     # ... of which 20 warm up (tools/benchmark.py:57-58)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 16; 32 for the fp8 R101 workload)")
    ap.add_argument("--workload", default="retinanet_r50_800x1344", choices=sorted(TRAIN_GFLOP_PER_IMG))
    ap.add_argument("--fp8", action="store_true",
                    help="fp8 (e4m3) weights and activations for the forward convolutions (BASELINE config 5: retinanet_r101_800x1344); "
                         "e5m2 data gradients under per-group delayed scales are on by default (--model-opt FP8_DGRAD=0: forward only)")
    ap.add_argument("--model-opt", action="append", default=[], metavar="KEY=VALUE",
                    help="ablation: set cfg.MODEL.KEY (e.g. FP8_1X1=0, FP8_DGRAD=0, FUSE_STEM_POOL=0); repeatable")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--ref-protocol-steps", type=int, default=10,
                    help="steps of the reference-protocol leg (host float64 batch in, device sync around every step); 0 = skip")
    ap.add_argument("--roofline-every", type=int, default=25,
                    help="record per-kernel HIP events on every n-th timed step (an event pair opens a ~10 us gap in the queue and the "
                         "instrumented step keeps the weight gradients on the main stream: ~2.7 ms per instrumented step; the headline "
                         "stays within ~1 %% of an uninstrumented run)")
    ap.add_argument("--dump-convs", action="store_true", help="per-shape kernel timing table on stderr")
    ap.add_argument("--seed", type=int, default=0, help="seed of the random-init weights (long-run stability studies)")
    ap.add_argument("--lr-scale", type=float, default=1.0, help="multiplies SOLVER.BASIC_LR (long-run stability studies)")
    ap.add_argument("--log-every", type=int, default=0,
                    help="long repeated-batch runs: print the loss (and the fp8 gradient-scale state) to stderr every n timed steps (a host sync each)")
    ap.add_argument("--serial-wgrad", action="store_true",
                    help="keep the weight-gradient kernels on the main stream for the whole run (what the instrumented steps do): "
                         "use it under rocprofv3 so that per-kernel durations are not inflated by concurrent kernels")
    ap.add_argument("--main-priority", type=int, default=None,
                    help="ablation: -1 = run the step's main chain on a high-priority stream owned by the solver (round 2's default); "
                         "0 / unset = the caller's stream")
    ap.add_argument("--roi-bwd-scatter", action="store_true",
                    help="Faster R-CNN: the general fp32 atomic scatter + conversion pass as RoIAlign backward (default: the tiled fixed-order sum)")
    ap.add_argument("--dense1x1", type=int, default=None, help="ablation: bd_conv_desc.route[0], the dense 1x1 mode of every call (0 = generic kernel for the dense 1x1 launches)")
    ap.add_argument("--no-mask-bits", action="store_true", help="ablation: bf16 activations instead of bit-packed ReLU gates as dgrad masks")
    ap.add_argument("--skip-s2-3x3-after-warmup", action="store_true",
                    help="timing A/B only: after the warm-up steps the 3x3 / stride-2 forward and data-gradient launches become no-ops (their "
                         "outputs keep the last warm-up step's values): the step-time difference is what those launches cost")
    ap.add_argument("--wgrad-knob", type=int, default=None,
                    help="ablation: bd_conv_desc.route[2], the weight-gradient bit mask (5 = the ring-staged weight-gradient kernels off: rounds 1-3 kernels)")
    ap.add_argument("--conv-knob", type=int, default=None,
                    help="ablation: bd_conv_desc.route[1] bit mask (include/basedet_hip.h) of every call of the run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)          # does not return
    worker(args)


def worker(args):
    global torch, np
    import numpy as _np
    import torch as _torch
    torch, np = _torch, _np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the basedet_amd path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    saved_stdout = _stdout_to_stderr()

    from basedet_amd import comm as bdcomm
    from basedet_amd import ops
    from basedet_amd.configs import (ATSSConfig, FasterRCNNConfig, FCOSConfig, FreeAnchorConfig, OTAConfig, RetinaNetConfig,
                                     retinanet_r18_config)
    from basedet_amd.models import ATSS, FCOS, OTA, FasterRCNN, FreeAnchor, RetinaNet, params as P
    from basedet_amd.solver import DetSolver, WarmupMultiStepLR, broadcast_parameters
    from basedet_amd.utils import DummyLoader

    comm = None
    if world > 1 or os.environ.get("BD_FORCE_ALLREDUCE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        comm = bdcomm.set_comm(bdcomm.Comm.from_env())       # bd_comm_init: ncclCommInitRank over the env:// store

    if args.batch is None:
        args.batch = 32 if args.fp8 else 16
    if args.workload == "retinanet_r50_800x1344":
        cfg, size = RetinaNetConfig(), (800, 1344)
    elif args.workload == "fcos_r50_800x1344":
        cfg, size = FCOSConfig(), (800, 1344)
    elif args.workload == "atss_r50_800x1344":
        cfg, size = ATSSConfig(), (800, 1344)
    elif args.workload == "freeanchor_r50_800x1344":
        cfg, size = FreeAnchorConfig(), (800, 1344)
    elif args.workload == "ota_r50_800x1344":
        cfg, size = OTAConfig(), (800, 1344)
    elif args.workload == "retinanet_r101_800x1344":
        cfg, size = RetinaNetConfig(), (800, 1344)
        cfg.MODEL.BACKBONE.NAME = "resnet101"
    elif args.workload == "faster_rcnn_r50_800x1344":
        cfg, size = FasterRCNNConfig(), (800, 1344)
    else:
        cfg, size = retinanet_r18_config(), (512, 512)
    cfg.MODEL.BATCHSIZE = args.batch
    if args.lr_scale != 1.0:
        cfg.SOLVER.BASIC_LR = cfg.SOLVER.BASIC_LR * args.lr_scale
    if args.fp8:
        cfg.MODEL.WEIGHT_DTYPE = "fp8_e4m3"
    for kv in args.model_opt:
        k, v = kv.split("=", 1)
        cfg.MODEL[k] = (float(v) if "." in v else int(v)) if v.lstrip("-").replace(".", "", 1).isdigit() else v
    # random-init weights of the named architecture; the last FrozenBN gamma of every residual branch is 0.2
    # (stand-in for ImageNet statistics: identity BN overflows a random ResNet-50; same FLOPs and bytes)
    if cfg.MODEL.NAME in ("FCOS", "ATSS", "OTA"):
        params = P.init_fcos_params(cfg, seed=args.seed, residual_gamma=0.2)
        model = {"FCOS": FCOS, "ATSS": ATSS, "OTA": OTA}[cfg.MODEL.NAME](cfg, params=params)
    elif cfg.MODEL.NAME == "FasterRCNN":
        params = P.init_faster_rcnn_params(cfg, seed=args.seed, residual_gamma=0.2)
        model = FasterRCNN(cfg, params=params)
    else:
        params = P.init_retinanet_params(cfg, seed=args.seed, residual_gamma=0.2)
        model = (FreeAnchor if cfg.MODEL.NAME == "FreeAnchor" else RetinaNet)(cfg, params=params)
    broadcast_parameters(model)
    solver = DetSolver.build(cfg, model)
    sched = WarmupMultiStepLR(solver.optimizer, cfg, world)     # LRSchedulerHook.before_iter (engine/hooks.py:218)

    loader = DummyLoader(args.batch, size, seed=rank)
    host_batch = next(loader)                                   # data float64 on the host, as the reference's loader yields it
    batch = {
        "data": torch.from_numpy(host_batch["data"].astype(np.float32)).cuda(),      # resident in HBM before the timed region
        "gt_boxes": torch.from_numpy(host_batch["gt_boxes"]).cuda(),
        "im_info": torch.from_numpy(host_batch["im_info"]).cuda(),
    }
    if os.environ.get("BD_PATCH3X3"):
        ops.set_route(patch3x3=int(os.environ["BD_PATCH3X3"]))
    timer = None if args.no_roofline else KernelTimer(ops)

    def sync():
        torch.cuda.synchronize()
        if comm is not None and world > 1:
            comm.barrier()

    last = None
    it = 0
    if args.serial_wgrad:
        model.async_wgrad = False
    if args.conv_knob is not None:
        ops.set_route(patch3x3=args.conv_knob)
    if args.dense1x1 is not None:
        ops.set_route(dense1x1=args.dense1x1)
    if args.wgrad_knob is not None:
        ops.set_route(wgrad=args.wgrad_knob)
    if args.no_mask_bits:
        model.use_mask_bits = False
    if args.roi_bwd_scatter:
        model.deterministic_roi_bwd = False
    if args.main_priority is not None:            # ablation: 0 = the step on the caller's (default-priority) stream
        solver.high_priority_main = args.main_priority < 0
    for _ in range(args.warmup):
        sched.step(it); it += 1
        last = solver.minimize(model, batch)
    sync()
    if args.skip_s2_3x3_after_warmup:             # (A/B only: see the flag's help; the result line says so)
        # (needs a -DBD_AB_SKIP diagnostic build of the library -- BD_LIB_NAME / BD_EXTRA_FLAGS, BASEDET_HIP_LIB: the shipped one answers the
        # first convolution call with BD_EINVAL)
        ops.set_route(patch3x3=(args.conv_knob if args.conv_knob is not None else 3) | 16)
    def _kernel_clock(kern, reset):
        """clock the chip held inside `kern` since the last reset (bd_probe_kernel_clock: workgroup 0 of every launch stamps its lifetime)"""
        import ctypes as _C
        mhz, busy = _C.c_double(0.0), _C.c_double(0.0)
        if ops.L().bd_probe_kernel_clock(kern.encode(), int(reset), _C.byref(mhz), _C.byref(busy)) != 0 or not mhz.value:
            return None
        return {"clock_mhz": round(mhz.value, 0), "probe_busy_ms": round(busy.value, 2)}
    CLOCKED = ("conv3x3_pp_kernel", "conv_wgrad3x3_ring_kernel")
    for kern_ in CLOCKED:
        _kernel_clock(kern_, True)               # (the device is idle: sync() above) the sums start with the timed region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]       # step boundaries on the main stream (p50 / p95)
    if comm is not None:
        solver.comm_profile = []          # per step: (backward done on the main stream, last bucket's all-reduce done on the comm stream)
    t0 = time.perf_counter()
    sampled = 0
    queue_mode = model.wgrad_queue_mode
    for k in range(args.steps):
        sched.step(it); it += 1
        if timer:
            timer.enabled = (k % max(1, args.roofline_every) == 0)
            sampled += int(timer.enabled)
            model.async_wgrad = not timer.enabled     # instrumented steps run serialised: clean per-kernel durations
            # ... and with one reduce per layer right behind its kernel, inside the event pair (the timed entry points are the un-queued ones)
            model.wgrad_queue_mode = "layer" if timer.enabled else queue_mode
        if args.serial_wgrad:
            model.async_wgrad = False
        marks[k].record()
        last = solver.minimize(model, batch)
        if args.log_every and (k + 1) % args.log_every == 0 and rank == 0:
            extra = ""
            if getattr(model, "fp8_group_scales", None):
                extra = " scales " + " ".join(f"{kk}:2^{int(np.log2(v))}" for kk, v in sorted(model.fp8_group_scales.items())[:8]) + \
                        " top(amax*scale) " + " ".join(f"{kk}:{v:.0f}" for kk, v in sorted(getattr(model, "fp8_last_fill", {}).items())[:8])
            print(f"# step {k + 1} loss {float(last['total_loss']):.4f}{extra}", file=sys.stderr, flush=True)
    marks[args.steps].record()
    sync()
    elapsed = time.perf_counter() - t0
    held_clock = {kern_: _kernel_clock(kern_, True) for kern_ in CLOCKED}          # over every launch of the timed region, all steps
    if timer:
        timer.enabled = False
        model.async_wgrad = not args.serial_wgrad
        model.wgrad_queue_mode = queue_mode
    if comm is not None and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        comm.allreduce(t, "max")
        elapsed = float(t.item())
    loss = float(last["total_loss"])
    assert np.isfinite(loss), "training diverged"
    step_ms = np.array([marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)])

    # ---- the reference harness's own protocol (tools/benchmark.py:125-133): host batch in, full device sync around every step
    ref_proto = None
    if args.ref_protocol_steps > 0:
        hb = {"data": host_batch["data"], "gt_boxes": host_batch["gt_boxes"], "im_info": host_batch["im_info"]}
        ts = []
        for k in range(args.ref_protocol_steps + 2):
            sched.step(it); it += 1
            torch.cuda.synchronize()
            a = time.perf_counter()
            solver.minimize(model, hb)
            torch.cuda.synchronize()
            if k >= 2:
                ts.append((time.perf_counter() - a) * 1e3)
        if comm is not None and world > 1:
            t = torch.tensor([float(np.mean(ts))], dtype=torch.float64, device="cuda")
            comm.allreduce(t, "max")
            mean_ms = float(t.item())
        else:
            mean_ms = float(np.mean(ts))
        ref_proto = {"images_per_sec": round(args.batch * world / (mean_ms * 1e-3), 2), "ms_per_step_mean": round(mean_ms, 3),
                     "ms_p50": round(float(np.percentile(ts, 50)), 3), "ms_p95": round(float(np.percentile(ts, 95)), 3),
                     "steps": len(ts),
                     "protocol": "basedet/tools/benchmark.py:125-133: float64 host batch -> fp32 -> H2D inside the step "
                                 "(pinned staging buffer), torch.cuda.synchronize() before and after every step; 2 untimed steps first"}

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / elapsed
        name = args.workload + ("_fp8w" if args.fp8 else "")
        out = {
            "metric": "images/sec training RetinaNet-R50-FPN 1333x800" if name == "retinanet_r50_800x1344" else "images/sec training " + name,
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if not args.fp8 else ("bf16 activations / fp8-e4m3 weights (forward%s), fp32 accumulate" % (" + e5m2 data gradients, per-group delayed scales" if cfg.MODEL.get("FP8_DGRAD", True) else "")),
            "data": "synthetic",
            "config": {"workload": f"{name} train step (fwd+bwd+allreduce+SGD), DummyLoader boxes, random-init weights, "
                                   "inputs resident in HBM",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "final_loss": round(loss, 4),
                       **({"fp8_grad_scale": [(int(a), float(c), float(d_)) for a, b_, c, d_ in (model.fp8_scale_log[:1] + model.fp8_scale_log[-2:])]}
                          if getattr(model, "fp8_scale_log", None) else {})},
            "step_ms_p50": round(float(np.percentile(step_ms, 50)), 3), "step_ms_p95": round(float(np.percentile(step_ms, 95)), 3),
            "step_ms_note": "device time between step boundaries on rank 0's main stream (HIP events), instrumented steps included",
        }
        if args.skip_s2_3x3_after_warmup:
            out["INVALID_AS_THROUGHPUT"] = "timing A/B: the 3x3 / stride-2 forward and data-gradient launches were skipped in the timed steps"
        if ref_proto:
            out["reference_protocol"] = ref_proto
        if comm is not None:
            # how much of the gradient exchange the backward pass did NOT hide: time from "backward (data + weight gradients) done on the
            # main stream" to "last bucket's all-reduce done on the communication stream", per timed step (<= 0: fully hidden)
            def _gap(a, b):
                try:
                    return a.elapsed_time(b)
                except RuntimeError:            # (a runtime that refuses stop-before-start pairs: the collectives finished first)
                    try:
                        return -b.elapsed_time(a)
                    except RuntimeError:
                        return 0.0
            ex = np.array([_gap(e0, e1) for e0, e1 in solver.comm_profile[:args.steps]] or [0.0])
            out["comm"] = {"rccl_ranks": world, "transport": "bd_comm_* (RCCL ncclAllReduce, one communicator, one high-priority stream)",
                           "buckets_bytes": {k: int((hi - lo) * 4) for k, (lo, hi) in sorted(solver.buckets.ranges.items(), key=lambda kv: -kv[1][0])},
                           "allreduce_exposed_ms": round(float(np.maximum(ex, 0.0).mean()), 3),
                           "allreduce_exposed_ms_p95": round(float(np.percentile(np.maximum(ex, 0.0), 95)), 3),
                           "allreduce_slack_ms_min": round(float(ex.min()), 3),
                           "note": "exposed = max(0, comm-stream done - main-stream backward done) per step, HIP events on rank 0"}
        gf = TRAIN_GFLOP_PER_IMG[args.workload]
        out["config"]["train_gflop_per_img"] = gf
        out["config"]["whole_step_mfma_frac"] = round(value / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4)
        if timer and args.dump_convs:
            timer.dump(sampled)
        if timer:
            entries = []
            # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the bench): valid for the
            # workload / batch AND the kernel build they were collected on (source digest recorded by scripts/pmc_traffic.py)
            pmc, pmc_note = {}, None
            pmc_path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
            if world == 1 and not args.no_pmc:
                pmc, pmc_note = measure_traffic(args)
            if pmc:
                pass
            elif name == "retinanet_r50_800x1344" and args.batch == 16 and os.path.exists(pmc_path):
                from basedet_amd import build as _b
                with open(pmc_path) as f:
                    rec = json.load(f)
                if rec.get("build_digest") == _b._digest():
                    pmc = rec.get("kernels", {})
                else:
                    pmc_note = "profiles/r03_pmc_traffic.json was collected on a different build of csrc/: traffic refused (null)"
            # What the matrix pipes of THIS device sustain on the kernels' register-level pattern (rotating random bf16 fragments, no LDS /
            # memory; bd_probe_mfma_rate, csrc/probe.hip) after 2 s of load: the chip holds ~2.1 GHz there, not the 2.4 GHz the vendor peak
            # assumes.  Reported beside `peak`, never instead of it.
            peak_meas = None
            if world == 1:
                import ctypes as _C
                tf_, clk_ = _C.c_double(0.0), _C.c_double(0.0)
                if ops.L().bd_probe_mfma_rate(2.0, _C.byref(tf_), _C.byref(clk_), ops.stream_ptr()) == 0:
                    peak_meas = {"tflops": round(tf_.value, 1), "clock_mhz": round(clk_.value, 0),
                                 "how": "bd_probe_mfma_rate: 8 x 4 rotating random bf16 fragments, 32 accumulators of v_mfma_f32_16x16x32_bf16 per wave, "
                                        "8 waves per CU, registers only, last 10 launches of a 2 s run"}
            for kern in timer.records:
                sm = timer.summary(kern)
                tf = sm["flops"] / (sm["ms"] * 1e-3) / 1e12
                gbs = sm["bytes"] / (sm["ms"] * 1e-3) / 1e9
                # which roof is lower for this kernel's mix of launches: arithmetic intensity against the machine balance
                mfma_peak = PEAK_FP8_TFLOPS if kern in ("conv_fp8_kernel", "conv3x3_pp8_kernel") else PEAK_BF16_TFLOPS      # e4m3 operands: ~5 PFLOP/s dense
                hbm_bound = sm["flops"] / sm["bytes"] < mfma_peak * 1e12 / (PEAK_HBM_GBS * 1e9)
                ach, peak, unit = (gbs, PEAK_HBM_GBS, "GB/s") if hbm_bound else (tf, mfma_peak, "TFLOP/s")
                entries.append({"bound": "hbm" if hbm_bound else "mfma", "kernel": kern, "achieved": round(ach, 2), "peak": peak,
                                "unit": unit, "frac": round(ach / peak, 4),
                                "traffic": pmc.get(kern, {}).get("hbm_bytes_per_launch"),
                                "tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1),
                                "algorithmic_bytes_per_launch": int(sm["bytes"] / sm["launches"]),
                                "launches_per_step": sm["launches"] // sampled,
                                "avg_launch_us": round(sm["ms"] * 1e3 / sm["launches"], 2),
                                "ms_per_step": round(sm["ms"] / sampled, 3),
                                "gflop_per_launch": round(sm["flops"] / sm["launches"] / 1e9, 2),
                                "sampled_steps": sampled})
                if peak_meas and not hbm_bound and mfma_peak == PEAK_BF16_TFLOPS:
                    entries[-1]["peak_measured"] = peak_meas["tflops"]
                    entries[-1]["frac_of_measured"] = round(tf / peak_meas["tflops"], 4)
                # frac = (in-cycle efficiency) x (held clock / 2400 MHz): the two factors, from the clock the chip held in this kernel's
                # launches of the PMC pass (GRBM_GUI_ACTIVE / 8 / dispatch time; the vendor peak is 2500 TFLOP/s at 2400 MHz)
                # ... and per LAUNCH CLASS: a kernel's average hides launches on the other side of the ridge
                cls = timer.by_launch_class(kern, mfma_peak * 1e12 / (PEAK_HBM_GBS * 1e9))
                if len(cls) > 1 or ("mfma" in cls) == hbm_bound:
                    entries[-1]["by_launch_class"] = {
                        c: {"launches_per_step": v["launches"] // sampled, "ms_per_step": round(v["ms"] / sampled, 3),
                            "flop_per_byte": round(v["flops"] / max(v["bytes"], 1.0), 1),
                            **({"achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
                                "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / mfma_peak, 4)} if c == "mfma" else
                               {"achieved": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "unit": "GB/s",
                                "frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})}
                        for c, v in cls.items()}
                clk_pmc = pmc.get(kern, {}).get("clock_mhz")
                held = held_clock.get(kern)
                if not hbm_bound and (held or clk_pmc):
                    if held:            # the kernel's own stamps over the WHOLE timed region (every step, concurrent side streams and all)
                        entries[-1]["clock_mhz"] = held["clock_mhz"]
                        entries[-1]["clock_src"] = ("in-kernel: workgroup 0 of every launch in the timed region adds d(s_memtime) and d(s_memrealtime) of its "
                                                    f"lifetime to two device counters (bd_probe_kernel_clock; {held['probe_busy_ms']} ms of workgroup lifetime)")
                    else:
                        entries[-1]["clock_mhz"] = clk_pmc
                        entries[-1]["clock_src"] = ("rocprofv3 --pmc GRBM_GUI_ACTIVE / 8 XCDs / dispatch time over this kernel's launches of the WRITE_SIZE pass "
                                                    f"(mean dispatch {pmc[kern].get('pmc_dispatch_us')} us; serialised launches, reads high on dispatches under 0.3 ms)")
                    if held and clk_pmc:
                        entries[-1]["clock_mhz_pmc"] = clk_pmc
                    # frac = frac_in_cycles x clock_mhz / 2400: what the kernel reaches of the matrix rate at the clock it was given
                    entries[-1]["frac_in_cycles"] = round(tf / (mfma_peak * entries[-1]["clock_mhz"] / 2400.0), 4)
            entries.sort(key=lambda e: -e["ms_per_step"])
            out["roofline"] = entries[0]                 # the dominant kernel of the step
            out["roofline_others"] = entries[1:]
            if pmc_note:
                out["roofline_traffic_note"] = pmc_note
            if peak_meas:
                out["mfma_peak_measured"] = peak_meas
        if world == 1 and not args.no_cpu_baseline and name == "retinanet_r50_800x1344":
            out["cpu_baseline"] = cpu_baseline(cfg, params)
        _print_result(saved_stdout, json.dumps(out))
    if comm is not None:
        comm.barrier()
        torch.cuda.synchronize()
        # every rank is past its last collective and its queue is empty: tear the communicator down in order (ncclCommDestroy +
        # store shutdown; round 3 left through os._exit here, which skipped both)
        comm.destroy()
        bdcomm.set_comm(None)


if __name__ == "__main__":
    main()
