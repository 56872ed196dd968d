"""DetTrainer (basedet/engine/trainer.py:12-100): the loop that calls the hot path -- `solver.minimize(model, batch)` once per
iteration (:98) with the LR schedule stepped before it (engine/hooks.py:218).  Everything else the reference hangs on its trainer
(checkpoint / eval / tensorboard hooks, EMA, resume) is control plane and out of this build's scope."""
import time

from . import comm as _comm
from .solver import WarmupMultiStepLR, clip_grad
from .utils.registry import registers


class Progress:
    """basecore.engine.Progress as the trainer uses it: 1-based epoch / iter counters over max_epoch x max_iter."""

    def __init__(self, max_epoch, max_iter):
        self.max_epoch, self.max_iter = max_epoch, max_iter
        self.epoch, self.iter = 1, 1

    @property
    def global_iter(self):
        return (self.epoch - 1) * self.max_iter + self.iter - 1


@registers.trainers.register()
class DetTrainer:
    def __init__(self, cfg, model, dataloader, solver, hooks=None):
        self.cfg, self.model, self.dataloader, self.solver = cfg, model, dataloader, solver
        self.dataloader_iter = iter(dataloader)
        s = cfg.SOLVER
        max_iter = int(s.NUM_IMAGE_PER_EPOCH / _comm.world_size() / cfg.MODEL.BATCHSIZE)          # trainer.py:45-48
        self.progress = Progress(s.MAX_EPOCH, max_iter)
        self.lr_scheduler = WarmupMultiStepLR(solver.optimizer, cfg, _comm.world_size())
        t = cfg.get("TRAINER", {})
        if t.get("AMP", {}).get("ENABLE", False):                              # trainer.py:52-54
            assert self.solver.grad_scaler is not None, "enable AMP but grad_scaler is None"
        gc = t.get("GRAD_CLIP", {})
        if gc.get("ENABLE", False):                                            # trainer.py:56-61
            self.solver.grad_clip_fn = clip_grad(self.model, gc["TYPE"], **dict(gc["ARGS"]))
        self.meter = {}
        self.log_interval = cfg.GLOBAL.LOG_INTERVAL
        self._hooks = list(hooks or [])

    def train_one_iter(self):
        """trainer.py:71-86."""
        t0 = time.time()
        inputs = next(self.dataloader_iter)
        t1 = time.time()
        loss_dict = self.model_step(inputs)
        self.meter = {name: float(v) for name, v in loss_dict.items()}        # float() syncs, like mge._full_sync() in the reference
        self.meter.update(train_time=time.time() - t1, data_time=t1 - t0)

    def model_step(self, model_inputs):
        """trainer.py:88-100."""
        return self.solver.minimize(self.model, model_inputs)

    def train(self, max_iters=None, log=print):
        """BaseTrainer.train: epochs x iterations; `max_iters` bounds a smoke run."""
        self.model.train()
        done = 0
        p = self.progress
        for p.epoch in range(1, p.max_epoch + 1):
            for p.iter in range(1, p.max_iter + 1):
                self.lr_scheduler.step(p.global_iter)                          # LRSchedulerHook.before_iter
                self.train_one_iter()
                done += 1
                if log and _comm.rank() == 0 and (done % self.log_interval == 0 or done == max_iters):
                    lr = self.solver.optimizer.param_groups[0]["lr"]
                    log(f"epoch {p.epoch}/{p.max_epoch} iter {p.iter}/{p.max_iter} lr {lr:.6f} " +
                        " ".join(f"{k} {v:.4f}" for k, v in self.meter.items()))
                if max_iters is not None and done >= max_iters:
                    return self.meter
        return self.meter
