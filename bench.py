#!/usr/bin/env python3
"""Headline benchmark: images/sec, training RetinaNet-R50-FPN on synthetic 1333x800 (padded 800x1344) images.

Protocol = the reference's own harness (basedet/tools/benchmark.py:125-140): device sync, K full training steps
(pre-process, forward, target assignment, losses, backward, gradient all-reduce, SGD + weight repack), device sync;
inputs are DummyLoader-shaped (basedet/utils/dummy.py:8-63) and already resident in HBM.  One process per GPU;
for N > 1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RCCL over xGMI).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the kernel with the largest share of the step (per-kernel entries: roofline_others).  Each conv kernel is priced
                  against the roof its launches sit under: "mfma" (algorithmic FLOPs / time vs 2.5 PFLOP/s) when their arithmetic
                  intensity exceeds the machine balance, else "hbm" (algorithmic bytes -- every operand once -- / time vs 8 TB/s);
                  durations are HIP events on the launch stream during the timed steps, `traffic` the PMC-measured HBM bytes
  cpu_baseline -- the CPU oracle's (oracle/model.py, torch-CPU fp32) training step on the host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

# HIP binds a stream to one of GPU_MAX_HW_QUEUES (default 4) hardware queues at its first use, sharing queues beyond that.  With
# the process group up, RCCL's and c10d's streams come first and the weight-gradient side stream ended up on the MAIN stream's
# queue: the two ran back to back (rocprofv3 kernel trace, Queue_Id column; scripts/queue_map.py).  Eight queues keep them apart.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's cross-process buffer sharing needs it on this driver

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense")
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


class ConvTimer:
    """HIP-event timing + algorithmic FLOP accounting of every conv launch (fwd / dgrad / wgrad)."""

    def __init__(self, ops):
        self.ops = ops
        self.records = {}
        self.enabled = False
        self.meta = {}
        self._orig = (ops.conv2d_fwd, ops.conv2d_dgrad, ops.conv2d_wgrad, ops.conv2d_wgrad_bias)

        def flops(d):
            m = sum(d.Ho[i] * d.Wo[i] for i in range(d.nseg)) * d.N
            return 2.0 * m * d.Cin * d.Cout * d.R * d.S

        def kernel_of(d, kind, dgrad=False):
            """Which HIP kernel serves this descriptor (mirrors the dispatch in csrc/conv_igemm.hip / conv3x3.hip /
            conv3x3_pp.hip / conv_wgrad.hip with the default bd_conv_set_patch3x3 mask)."""
            same = all(d.Hi[i] == d.Ho[i] and d.Wi[i] == d.Wo[i] for i in range(d.nseg))
            is3 = d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1 and same
            is1 = d.R == 1 and d.S == 1 and d.pad == 0
            if kind == "igemm":
                if not is3:
                    # generic kernel: BK = 32 instance for 1x1 filters (and Cin <= 32) and for stride-2 launches of >= 512 tiles,
                    # BK = 64 for the rest
                    ck_, co_ = (d.Cout, d.Cin) if dgrad else (d.Cin, d.Cout)
                    pix = d.N * sum((d.Hi[i] * d.Wi[i]) if dgrad else (d.Ho[i] * d.Wo[i]) for i in range(d.nseg))
                    big = d.stride == 2 and -(-pix // 128) * -(-co_ // 128) >= 512
                    return "conv_igemm_kernel<32>" if (d.R * d.S == 1 or ck_ <= 32 or big) else "conv_igemm_kernel<64>"
                ck, co = (d.Cout, d.Cin) if dgrad else (d.Cin, d.Cout)
                pp = ck % 8 == 0 and co > 128 and co % 8 == 0
                if pp:
                    return "conv3x3_pp_kernel"
                # Cout <= 64: the 64-channel tile of conv3x3_pp128.hip; the rest: conv3x3.hip
                return "conv3x3_pp128_kernel" if (ck % 8 == 0 and co % 8 == 0 and co <= 64) else "conv3x3_patch_kernel"
            is3w = d.R == 3 and d.S == 3 and d.pad == 1 and all((d.Hi[i] - 1) // d.stride + 1 == d.Ho[i] for i in range(d.nseg))
            return "conv_wgrad3x3_kernel" if is3w else ("conv_wgrad1x1_kernel" if is1 else "conv_wgrad_kernel")

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
                kern = kernel_of(d, kind, fn.__name__ == "conv2d_dgrad")
                self.records.setdefault(kern, []).append((s, e, flops(d)))
                # algorithmic HBM bytes: every operand once -- both activations, the weights, and the epilogue's add / mask
                # operands (residual, ReLU mask), which have the shape of the result
                mi = sum(d.Hi[i] * d.Wi[i] for i in range(d.nseg)) * d.N
                mo = sum(d.Ho[i] * d.Wo[i] for i in range(d.nseg)) * d.N
                nbytes = 2.0 * (mi * d.Cin + mo * d.Cout) + (4.0 if kind == "wgrad" else 2.0) * d.Cin * d.Cout * d.R * d.S
                if kind == "igemm":
                    res = 2.0 * (mi * d.Cin if fn.__name__ == "conv2d_dgrad" else mo * d.Cout)
                    nbytes += res * ((k.get("add") is not None) + (k.get("mask") is not None))
                self.meta.setdefault(kern, []).append((fn.__name__, d.Cin, d.Cout, d.R, d.stride, d.nseg, d.Ho[0], d.Wo[0], nbytes))
                return r
            return inner

        ops.conv2d_fwd = wrap(ops.conv2d_fwd, "igemm")
        ops.conv2d_dgrad = wrap(ops.conv2d_dgrad, "igemm")
        ops.conv2d_wgrad = wrap(ops.conv2d_wgrad, "wgrad")
        ops.conv2d_wgrad_bias = wrap(ops.conv2d_wgrad_bias, "wgrad")

    def dump(self, steps):
        agg = {}
        for kind, rec in self.records.items():
            for (s, e, f), m in zip(rec, self.meta.get(kind, [])):
                k = (kind,) + m[:8]
                a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += s.elapsed_time(e); a[2] += f; a[3] += m[8]
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        print("# per-shape conv time per step (ms), TFLOP/s", file=sys.stderr)
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


def cpu_baseline(cfg, params, seconds=20.0, batch=2, size=(800, 1344)):
    """Oracle training step (fwd + bwd + SGD) on the host cores, bounded sample of the same workload."""
    from basedet_amd.models import params as P
    from basedet_amd.utils import DummyLoader
    from oracle.model import Oracle
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    b = next(DummyLoader(batch, size, seed=0))
    b["data"] = b["data"].astype(np.float32)
    state = {}
    lr = cfg.SOLVER.BASIC_LR * batch
    t_all, iters = [], 0
    t_start = time.time()
    while True:
        t0 = time.time()
        losses, _ = orc.retinanet_losses(b)
        g = orc.grads(losses["total_loss"])
        orc.sgd_step(g, state, lr, 0.9, cfg.SOLVER.WEIGHT_DECAY)
        dt = time.time() - t0
        iters += 1
        if iters > 1:
            t_all.append(dt)          # first iteration = warm-up
        if (time.time() - t_start > seconds and t_all) or iters >= 4:
            break
    mean = float(np.mean(t_all))
    return dict(value=batch / mean, unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/model.py torch-CPU fp32 RetinaNet-R50 step, batch {batch} x {size[0]}x{size[1]}, "
                       f"{len(t_all)} timed iteration(s) after 1 warm-up, {mean:.2f} s/iter")


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--workload", default="retinanet_r50_800x1344", choices=sorted(TRAIN_GFLOP_PER_IMG))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-every", type=int, default=10,
                    help="record per-conv HIP events on every n-th timed step (an event pair opens a ~10 us gap in the queue and the "
                         "instrumented step keeps the weight gradients on the main stream: ~2.7 ms per instrumented step; with the "
                         "default 10 steps one step is sampled and the headline stays within ~1 %% of an uninstrumented run)")
    ap.add_argument("--dump-convs", action="store_true", help="per-shape conv timing table on stderr")
    ap.add_argument("--serial-wgrad", action="store_true",
                    help="keep the weight-gradient kernels on the main stream for the whole run (what the instrumented steps do): "
                         "use it under rocprofv3 so that per-kernel durations are not inflated by concurrent kernels")
    ap.add_argument("--conv-knob", type=int, default=None,
                    help="ablation: bd_conv_set_patch3x3 bit mask (include/basedet_hip.h) applied before the run")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the basedet_amd path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    saved_stdout = _stdout_to_stderr()
    import torch.distributed as dist
    if world > 1 or os.environ.get("BD_FORCE_ALLREDUCE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from basedet_amd import ops
    from basedet_amd.configs import (ATSSConfig, FasterRCNNConfig, FCOSConfig, FreeAnchorConfig, OTAConfig, RetinaNetConfig,
                                     retinanet_r18_config)
    from basedet_amd.models import ATSS, FCOS, OTA, FasterRCNN, FreeAnchor, RetinaNet, params as P
    from basedet_amd.solver import DetSolver, WarmupMultiStepLR, broadcast_parameters
    from basedet_amd.utils import DummyLoader

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
    # random-init weights of the named architecture; the last FrozenBN gamma of every residual branch is 0.2
    # (stand-in for ImageNet statistics: identity BN overflows a random ResNet-50; same FLOPs and bytes)
    if cfg.MODEL.NAME in ("FCOS", "ATSS", "OTA"):
        params = P.init_fcos_params(cfg, seed=0, residual_gamma=0.2)
        model = {"FCOS": FCOS, "ATSS": ATSS, "OTA": OTA}[cfg.MODEL.NAME](cfg, params=params)
    elif cfg.MODEL.NAME == "FasterRCNN":
        params = P.init_faster_rcnn_params(cfg, seed=0, residual_gamma=0.2)
        model = FasterRCNN(cfg, params=params)
    else:
        params = P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2)
        model = (FreeAnchor if cfg.MODEL.NAME == "FreeAnchor" else RetinaNet)(cfg, params=params)
    broadcast_parameters(model)
    solver = DetSolver.build(cfg, model)
    sched = WarmupMultiStepLR(solver.optimizer, cfg, world)     # LRSchedulerHook.before_iter (engine/hooks.py:218)

    loader = DummyLoader(args.batch, size, seed=rank)
    b = next(loader)
    batch = {
        "data": torch.from_numpy(b["data"].astype(np.float32)).cuda(),      # resident in HBM before the timed region
        "gt_boxes": torch.from_numpy(b["gt_boxes"]).cuda(),
        "im_info": torch.from_numpy(b["im_info"]).cuda(),
    }
    if os.environ.get("BD_PATCH3X3"):
        ops.L().bd_conv_set_patch3x3(int(os.environ["BD_PATCH3X3"]))
    timer = None if args.no_roofline else ConvTimer(ops)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    last = None
    it = 0
    if args.serial_wgrad:
        model.async_wgrad = False
    if args.conv_knob is not None:
        from basedet_amd import ops as _ops
        _ops.L().bd_conv_set_patch3x3(args.conv_knob)
    for _ in range(args.warmup):
        sched.step(it); it += 1
        last = solver.minimize(model, batch)
    sync()
    t0 = time.perf_counter()
    sampled = 0
    for k in range(args.steps):
        sched.step(it); it += 1
        if timer:
            timer.enabled = (k % max(1, args.roofline_every) == 0)
            sampled += int(timer.enabled)
            model.async_wgrad = not timer.enabled     # instrumented steps run serialised: clean per-kernel durations
        if args.serial_wgrad:
            model.async_wgrad = False
        last = solver.minimize(model, batch)
    sync()
    elapsed = time.perf_counter() - t0
    if timer:
        timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(last["total_loss"])
    assert np.isfinite(loss), "training diverged"

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / elapsed
        out = {
            "metric": "images/sec training RetinaNet-R50-FPN 1333x800" if args.workload == "retinanet_r50_800x1344" else "images/sec training " + args.workload,
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload} train step (fwd+bwd+allreduce+SGD), DummyLoader boxes, random-init weights",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "final_loss": round(loss, 4)},
        }
        gf = TRAIN_GFLOP_PER_IMG[args.workload]
        out["config"]["train_gflop_per_img"] = gf
        out["config"]["whole_step_mfma_frac"] = round(value / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4)
        if timer and args.dump_convs:
            timer.dump(sampled)
        if timer:
            entries = []
            # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the bench):
            # valid for the workload / batch they were collected on, null otherwise
            pmc = {}
            pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if args.workload == "retinanet_r50_800x1344" and args.batch == 16 and os.path.exists(pmc_path):
                with open(pmc_path) as f:
                    pmc = json.load(f).get("kernels", {})
            for kern in timer.records:
                sm = timer.summary(kern)
                tf = sm["flops"] / (sm["ms"] * 1e-3) / 1e12
                gbs = sm["bytes"] / (sm["ms"] * 1e-3) / 1e9
                # which roof is lower for this kernel's mix of launches: arithmetic intensity against the machine balance
                hbm_bound = sm["flops"] / sm["bytes"] < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
                ach, peak, unit = (gbs, PEAK_HBM_GBS, "GB/s") if hbm_bound else (tf, PEAK_BF16_TFLOPS, "TFLOP/s")
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
            entries.sort(key=lambda e: -e["ms_per_step"])
            out["roofline"] = entries[0]                 # the dominant kernel of the step
            out["roofline_others"] = entries[1:]
        if world == 1 and not args.no_cpu_baseline and args.workload == "retinanet_r50_800x1344":
            out["cpu_baseline"] = cpu_baseline(cfg, params)
        _print_result(saved_stdout, json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        torch.cuda.synchronize()
        # every rank is past its last collective: leave without the process-group teardown (c10d's watchdog / heartbeat threads can
        # hold a finished process for minutes -- one GPU test run here took 10 min instead of 1 with all tests passing)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
