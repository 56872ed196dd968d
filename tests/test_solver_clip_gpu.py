"""Gradient clipping on the HIP path (engine/trainer.py:57-61 -> basecore clip_grad -> megengine.optimizer.clip_grad_value / clip_grad_norm;
configs/extra_cfg.py:99-105) and the AMP protocol check (engine/trainer.py:52-54).  The oracle here is the formula itself in float64 on
the same gradient arena; the update is then checked through the SGD launch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_clip_kernels_match_float64_formula():
    from basedet_amd import ops
    rng = np.random.default_rng(0)
    for n in (1, 5, 1023, 1 << 20, 3_000_001):
        g0 = (rng.standard_normal(n) * rng.choice([1e-3, 1.0, 30.0], n)).astype(np.float32)
        for pre in (1.0, 0.125):
            g = torch.from_numpy(g0).cuda()
            ops.clip_grad_value(g, -0.5, 0.75, pre)
            want = np.clip(g0 * np.float32(pre), -0.5, 0.75)
            assert np.array_equal(g.cpu().numpy(), want)
            for ordv, max_norm in ((2.0, 1.0), (2.0, 1e9), (float("inf"), 0.5), (1.0, 10.0), (3.0, 2.0)):
                g = torch.from_numpy(g0).cuda()
                out = torch.zeros((1,), dtype=torch.float32, device="cuda")
                ops.clip_grad_norm(g, max_norm, ordv, pre, out)
                a = np.abs(g0.astype(np.float64)) * pre
                nrm = a.max() if np.isinf(ordv) else (a ** ordv).sum() ** (1.0 / ordv)
                got_n = float(out.item())
                assert abs(got_n - nrm) <= 2e-6 * nrm, (n, ordv, got_n, nrm)
                f = np.float32(pre) * np.float32(min(np.float32(max_norm) / (np.float32(got_n) + np.float32(1e-6)), 1.0))
                want = g0 * f
                assert np.allclose(g.cpu().numpy(), want, rtol=2e-7, atol=0), (n, ordv)
                if max_norm == 1e9 and pre == 1.0:
                    assert np.array_equal(g.cpu().numpy(), g0)          # no clipping: scale is exactly 1
    # bitwise reproducible
    g0 = rng.standard_normal(3_000_001).astype(np.float32)
    outs = []
    for _ in range(2):
        g = torch.from_numpy(g0).cuda()
        ops.clip_grad_norm(g, 1.0, 2.0, 1.0)
        outs.append(g.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("ctype,args", [("value", dict(lower=-1e-3, upper=1e-3)), ("norm", dict(max_norm=0.5, ord=2))])
def test_trainer_installs_and_solver_applies_grad_clip(ctype, args):
    from basedet_amd.engine import DetTrainer
    from basedet_amd.models import RetinaNet
    from basedet_amd.solver import DetSolver, GradClip
    from tests.test_model_gpu import _setup
    cfg, params, batch = _setup("resnet18", 2, (128, 160))
    cfg.TRAINER.GRAD_CLIP = dict(ENABLE=True, TYPE=ctype, ARGS=args)
    cfg.TRAINER.AMP.ENABLE = True
    model = RetinaNet(cfg, params=params)
    solver = DetSolver.build(cfg, model)
    assert solver.grad_scaler is not None and solver.grad_scaler.scale_factor == 128.0      # default_solver.py:66-76 (static scale)
    solver.optimizer.param_groups[0]["lr"] = 0.1         # updates well above the fp32 resolution of the weights
    trainer = DetTrainer(cfg, model, iter([batch]), solver)
    assert isinstance(solver.grad_clip_fn, GradClip)
    # reference run without clipping: same forward / backward, gradients kept
    model(batch)
    model.backward()
    torch.cuda.synchronize()
    g_ref = model.arena.g.double().cpu().numpy().copy()
    w0 = model.arena.w.double().cpu().numpy().copy()
    assert not model.arena.v.any()
    trainer.model_step(batch)
    torch.cuda.synchronize()
    if ctype == "value":
        g_want = np.clip(g_ref, args["lower"], args["upper"])
        assert (np.abs(g_ref) > args["upper"]).any()            # the clip is active on this model
    else:
        nrm = np.sqrt((g_ref ** 2).sum())
        assert nrm > args["max_norm"]
        g_want = g_ref * min(1.0, args["max_norm"] / (nrm + 1e-6))
        assert abs(float(solver.grad_clip_fn.last_norm.item()) - nrm) < 1e-4 * nrm
    lr, wd = solver.optimizer.param_groups[0]["lr"], solver.optimizer.param_groups[0]["weight_decay"]
    w_want = w0 - lr * (g_want + wd * w0)
    got = model.arena.w.double().cpu().numpy()
    # w is fp32: the update (lr * g ~ 1e-6) is resolved to half an ulp of w
    assert np.allclose(got - w0, w_want - w0, rtol=1e-4, atol=1.2e-7 * np.abs(w0).max())
    big = np.abs(w_want - w0) > 1e-6 * np.abs(w0).max()              # where the update is well above w's rounding: relative check
    assert big.any() and np.allclose((got - w0)[big], (w_want - w0)[big], rtol=0.2)
