"""DetSolver (basedet/solver/default_solver.py:79-124) + the basecore `Solver.minimize` step it returns.

lr = BASIC_LR * BATCHSIZE * world_size for MEAN reduction (:99-106); SGD(momentum, weight_decay) over the
trainable parameter arena in ONE fused launch; gradients are all-reduced (mean) over RCCL in arena buckets that
follow the backward order (head -> FPN -> layer4 -> ... -> layer2) on the communicator's stream, overlapped with backward.
The collectives are the bd_comm_* entry points of the C ABI (basedet_amd/comm.py); no c10d process group is involved.
"""
import torch

from .. import comm as _comm
from .. import ops
from ..utils.registry import registers


class _ParamGroupView(dict):
    pass


class SGD:
    """megengine.optimizer.SGD semantics: g' = g + wd*w; v = momentum*v + g'; w -= lr*v (fused HIP kernel)."""

    def __init__(self, model, lr, weight_decay, momentum=0.9):
        self.model = model
        self.param_groups = [_ParamGroupView(lr=lr, weight_decay=weight_decay, momentum=momentum)]

    def step(self, grad_scale=1.0):
        a = self.model.arena
        g = self.param_groups[0]
        ops.sgd_momentum_step(a.w, a.v, a.g, g["lr"], g["momentum"], g["weight_decay"], grad_scale)
        self.model.repack_trainable()
        return self

    def clear_grad(self):
        return self   # every gradient slot is overwritten by the next backward (no accumulation across steps)


class GradBuckets:
    """Asynchronous bucketed all-reduce of the gradient arena (replaces dist.make_allreduce_cb,
    solver/default_solver.py:121).  Buckets = contiguous arena ranges closed in backward order."""

    def __init__(self, model, mode="MEAN", comm=None, wire_dtype="fp32"):
        self.model = model
        self.mode = mode
        # SOLVER.ALLREDUCE_DTYPE = "bf16" (opt-in): the buckets cross xGMI as bf16 (75 MB instead of 151 MB per step for RetinaNet-R50);
        # the update then differs from the fp32 exchange by bf16 resolution (rel-L2 ~1e-3: tests/test_dist_cpu.py, test_dist_gpu.py)
        assert wire_dtype in ("fp32", "bf16"), wire_dtype
        self.wire_dtype = wire_dtype
        self._wire_tmp = None
        self.comm = comm if comm is not None else _comm.get_comm()
        # a communicator of one rank (BD_FORCE_ALLREDUCE=1 in bench / tests) still sends every bucket through RCCL: the identity
        # reduction exercises the stream plumbing on a single GPU
        self.enabled = self.comm is not None
        self.world = self.comm.world if self.enabled else 1
        self.ranges = self._ranges()

    def _ranges(self):
        """name of the backward phase -> (start, end) element range of the arena."""
        ent = self.model.arena.entries
        groups = {}
        for name, _, off, n in ent:
            if name.startswith(("head.", "rpn.", "rcnn.")):
                k = "head"
            elif "fpn_" in name or "top_block" in name:
                k = "fpn"
            else:
                k = name.split(".")[2]     # backbone.bottom_up.layerX...
            lo, hi = groups.get(k, (off, off))
            groups[k] = (min(lo, off), max(hi, (off + n + 63) // 64 * 64))
        return groups

    def on_ready(self, phase, producers=()):
        """All gradients of `phase` have been ENQUEUED on the current stream and on the `producers` streams (the weight-gradient
        side stream).  bd_comm_allreduce_async records an event on each of them, makes the communication stream wait for those
        events and enqueues the collective there -- the compute streams never wait for each other or for the collective, and the
        host does not block."""
        if not self.enabled or phase not in self.ranges:
            return
        lo, hi = self.ranges[phase]
        buf = self.model.arena.g[lo:hi]
        prod = []
        if buf.is_cuda:
            prod = [torch.cuda.current_stream()] + [s for s in producers if s is not None]
        if self.wire_dtype == "bf16":
            if self._wire_tmp is None:
                self._wire_tmp = torch.empty(self.model.arena.g.numel(), dtype=torch.bfloat16, device=buf.device)
            self.comm.allreduce_async_bf16(buf, self._wire_tmp[lo:hi], prod, "sum")      # each bucket has its own slice of the scratch
        else:
            self.comm.allreduce_async(buf, prod, "sum")

    def wait(self):
        """The current stream (the SGD launch comes next) waits for every bucket; returns the gradient scale of the reduce mode."""
        if self.enabled:
            self.comm.wait()
        return 1.0 / self.world if (self.enabled and self.mode == "MEAN") else 1.0


class GradClip:
    """What basecore's `clip_grad(params, type, **args)` returns (engine/trainer.py:57-61; configs/extra_cfg.py:99-105): a callable the
    solver invokes between the gradient all-reduce and the optimizer step.  TYPE "value" (ARGS lower, upper) =
    megengine.optimizer.clip_grad_value, TYPE "norm" (ARGS max_norm, ord = 2) = clip_grad_norm over ALL trainable gradients -- here one
    or three launches over the flat gradient arena (bd_clip_grad_value / bd_clip_grad_norm), no host synchronisation.  The
    reduce-mode factor (1 / world) the SGD launch would apply is folded in (`pre_scale`): the reference clips the averaged gradients."""

    def __init__(self, model, clip_type="value", **args):
        assert clip_type in ("value", "norm"), clip_type            # basecore asserts the same two types
        self.model, self.type, self.args = model, clip_type, dict(args)
        if clip_type == "value":
            self.lower, self.upper = float(args["lower"]), float(args["upper"])
        else:
            self.max_norm, self.ord = float(args["max_norm"]), float(args.get("ord", 2.0))
        self.last_norm = None          # device scalar: the gradient norm of the last step (TYPE "norm")
        self._ws = None

    def __call__(self, pre_scale=1.0):
        g = self.model.arena.g
        if self.type == "value":
            ops.clip_grad_value(g, self.lower, self.upper, pre_scale)
        else:
            if self._ws is None:
                self._ws = torch.empty((ops.clip_grad_norm_workspace_bytes(),), dtype=torch.uint8, device=g.device)
                self.last_norm = torch.zeros((1,), dtype=torch.float32, device=g.device)
            ops.clip_grad_norm(g, self.max_norm, self.ord, pre_scale, self.last_norm, self._ws)
        return 1.0


def clip_grad(model, clip_type="value", **args):
    """basecore.engine.clip_grad with the model in place of its parameter list (the gradients live in model.arena.g)."""
    return GradClip(model, clip_type, **args)


class GradScaler:
    """Stand-in for megengine.amp.GradScaler as DetSolver builds it (solver/default_solver.py:66-76: init_scale 65536 with
    DYNAMIC_SCALE, else 128; growth_interval 2000 / 0).  The reference needs it because its AMP is fp16; this build's mixed precision is
    bf16 activations with fp32 accumulation and fp32 master weights, which has fp32's exponent range, so the scale factor is
    carried for the protocol (`solver.grad_scaler is not None`, engine/trainer.py:53-54) and never applied."""

    def __init__(self, init_scale=128.0, growth_interval=0):
        self.scale_factor = float(init_scale)
        self.growth_interval = int(growth_interval)


class Solver:
    """basecore.engine.Solver protocol used by DetTrainer (engine/trainer.py:55-61,98)."""

    def __init__(self, optimizer, buckets, grad_scaler=None, grad_clip_fn=None):
        self.optimizer = optimizer
        self.buckets = buckets
        self.grad_scaler = grad_scaler
        self.grad_clip_fn = grad_clip_fn

    def minimize(self, model, inputs):
        """One training step (basecore Solver.minimize; called at engine/trainer.py:98).  `high_priority_main = True` runs the step's main
        chain (forward, losses, data gradients, SGD) on a HIGH-priority stream owned by the solver, ahead of the weight-gradient side
        stream at workgroup dispatch (the caller's stream waits for it at the end).  Round 2 measured +0.8 % for it; since layer1 runs as
        one persistent launch per block (round 3) the caller's default-priority stream is the faster one -- 9 of 9 alternations on two
        boxes, +0.5 % on average -- and is the default."""
        hp = self._main_stream(model)
        if hp is None:
            return self._step(model, inputs)
        cur = torch.cuda.current_stream()
        hp.wait_stream(cur)
        with torch.cuda.stream(hp):
            losses = self._step(model, inputs)
        cur.wait_stream(hp)
        return losses

    def _main_stream(self, model):
        if not getattr(self, "high_priority_main", False) or not torch.cuda.is_available() or getattr(model, "device", None) is None \
                or model.device.type != "cuda":
            return None
        if getattr(self, "_hp", None) is None:
            self._hp = torch.cuda.Stream(device=model.device, priority=-1)
        return self._hp

    def _step(self, model, inputs):
        losses = model(inputs)
        model.backward(on_bucket_ready=self.buckets.on_ready)
        prof = getattr(self, "comm_profile", None)        # bench.py (N > 1): [(backward done on this stream, collectives done on the comm stream)]
        if prof is not None and self.buckets.enabled and torch.cuda.is_available():
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        scale = self.buckets.wait()
        if prof is not None and self.buckets.enabled and torch.cuda.is_available():
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(self.buckets.comm.stream)
            prof.append((e0, e1))
        if self.grad_clip_fn is not None:                 # engine/trainer.py:57-61: between the all-reduce and the optimizer step
            if isinstance(self.grad_clip_fn, GradClip):
                scale = self.grad_clip_fn(pre_scale=scale)
            else:                                         # a foreign callable sees the averaged gradients
                if scale != 1.0:
                    model.arena.g.mul_(scale)
                self.grad_clip_fn()
                scale = 1.0
        self.optimizer.step(grad_scale=scale)
        self.optimizer.clear_grad()
        return losses


class WarmupMultiStepLR:
    """LRSchedulerHook.build_lr_scheduler (engine/hooks.py:222-248): MultiStepLR with iteration-wise milestones
    (epochs * iters_per_epoch) wrapped in basecore's WarmUpScheduler(scheduler, warmup_length=WARM_ITERS).

    basecore is not vendored, so the SHAPE of its warm-up ramp is an assumption of this build (parity unpinned):
    by default a linear ramp lr * (it + 1) / WARM_ITERS that reaches the base rate at the last warm-up iteration.  Two config keys
    select the other common forms should basecore's differ: SOLVER.WARMUP_START_FACTOR = f ramps lr * (f + (1 - f) * it / WARM_ITERS)
    (detectron-style, starts at f * lr); SOLVER.WARMUP_MODE = "constant" holds lr * f for the whole warm-up."""

    def __init__(self, optimizer, cfg, world_size=1):
        s = cfg.SOLVER
        self.optimizer = optimizer
        self.base_lr = optimizer.param_groups[0]["lr"]
        iters_per_epoch = int(s.NUM_IMAGE_PER_EPOCH / world_size / cfg.MODEL.BATCHSIZE)     # engine/trainer.py:48
        self.milestones = [int(e) * iters_per_epoch for e in s.LR_DECAY_STAGES]
        self.gamma = s.LR_DECAY_RATE
        self.warm_iters = s.get("WARM_ITERS", 0)
        self.warm_mode = s.get("WARMUP_MODE", "linear")
        self.warm_start = s.get("WARMUP_START_FACTOR", None)
        assert self.warm_mode in ("linear", "constant")

    def lr_at(self, it):
        lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if it >= m)
        if it < self.warm_iters:
            if self.warm_mode == "constant":
                lr *= self.warm_start if self.warm_start is not None else 1.0 / self.warm_iters
            elif self.warm_start is not None:
                lr *= self.warm_start + (1.0 - self.warm_start) * it / self.warm_iters
            else:
                lr *= (it + 1) / self.warm_iters
        return lr

    def step(self, it):
        self.optimizer.param_groups[0]["lr"] = self.lr_at(it)


@registers.solvers.register()
class DetSolver:
    @classmethod
    def build(cls, cfg, model):
        solver_cfg = cfg.SOLVER
        mode = solver_cfg.get("REDUCE_MODE", "MEAN")
        assert mode in ["MEAN", "SUM"]
        world = _comm.world_size()
        lr = solver_cfg.BASIC_LR * cfg.MODEL.BATCHSIZE
        wd = solver_cfg.WEIGHT_DECAY
        if mode == "MEAN":
            lr = lr * world
        else:
            wd = wd * world
        extra = dict(solver_cfg.get("EXTRA_OPT_ARGS", {}))
        opt = SGD(model, lr=lr, weight_decay=wd, momentum=extra.get("momentum", 0.0))
        amp = cfg.get("TRAINER", {}).get("AMP", {}) if hasattr(cfg, "get") else {}
        scaler = None
        if amp.get("ENABLE", False):                      # default_solver.py:66-76
            dyn = bool(amp.get("DYNAMIC_SCALE", False))
            scaler = GradScaler(65536.0 if dyn else 128.0, 2000 if dyn else 0)
        return Solver(opt, GradBuckets(model, mode, wire_dtype=str(solver_cfg.get("ALLREDUCE_DTYPE", "fp32"))), grad_scaler=scaler)


def broadcast_parameters(model, src=0):
    """configs/detection_cfg.py:80-82 (dist.bcast_list_ of params and buffers) as one flat broadcast."""
    c = _comm.get_comm()
    if c is not None and c.world > 1:
        c.bcast(model.arena.w, root=src)
        model.repack_trainable()
