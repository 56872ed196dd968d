"""BASELINE config 5 on its own backbone: RetinaNet-R101-FPN (models/cls/resnet.py:289-293: Bottleneck, [3, 4, 23, 3]) at 2 x 800 x 1344,
bf16 and WEIGHT_DTYPE = fp8_e4m3 (the reference's mixed-precision hook is fp16 autocast, solver/default_solver.py:66-76; fp8 has no
reference counterpart, so its tolerances are stated against fp32 and against the bf16 path).

bf16:  labels bit-exact vs oracle.model, losses / logits <= 2e-2, every parameter gradient <= 2e-2 rel-L2 against the oracle evaluated on
       the HIP run's stored activations (identical ReLU gates) and within the loose bound against the plain fp32 oracle.
fp8:   labels bit-exact, losses <= 5e-2 vs the fp32 oracle, gradient cosine >= 0.98 vs the bf16 path, for the forward-only mode and for
       the e5m2 data-gradient mode (per-group delayed scales: the default of config 5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZE = (800, 1344)
N = 2


def _setup():
    from tests.test_model_gpu import _setup as base
    from basedet_amd.configs import RetinaNetConfig
    from basedet_amd.models import params as P
    cfg, _, batch = base("resnet50", N, SIZE)
    cfg = RetinaNetConfig()
    cfg.MODEL.BATCHSIZE = N
    cfg.MODEL.BACKBONE.NAME = "resnet101"
    params = P.init_retinanet_params(cfg, 0)
    rng = np.random.default_rng(1)
    for k in list(params):                       # non-trivial FrozenBN statistics; the last BN of every branch damped (33 blocks deep)
        if k.endswith("running_var"):
            params[k] = rng.uniform(0.5, 1.5, params[k].shape).astype(np.float32)
        elif k.endswith("running_mean"):
            params[k] = rng.normal(0, 0.1, params[k].shape).astype(np.float32)
        elif (".bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            lo, hi = (0.1, 0.25) if ".bn3." in k else (0.7, 1.3)
            params[k] = rng.uniform(lo, hi, params[k].shape).astype(np.float32)
        elif (".bn" in k or "downsample.1" in k) and k.endswith(".bias"):
            params[k] = rng.normal(0, 0.1, params[k].shape).astype(np.float32)
    return cfg, params, batch


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def r101():
    """One oracle evaluation and one bf16 HIP step shared by the tests of this module (the oracle step takes ~30 s of host time)."""
    from basedet_amd.models import RetinaNet, params as P
    from oracle.model import Oracle
    cfg, params, batch = _setup()
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    ref, aux = orc.retinanet_losses(batch)
    ref_grads = {k: v.detach().clone() for k, v in orc.grads(ref["total_loss"]).items()}
    ref = {k: float(v.detach()) for k, v in ref.items()}
    aux = dict(labels=aux["labels"], num_fg=aux["num_fg"], logits=aux["logits"].detach())
    model = RetinaNet(cfg, params=params)
    assert len(model.blocks) == 33 and sum(1 for b in model.blocks if b["layer"] == 3) == 23
    out = model(batch)
    pl = model._cur
    res = dict(labels=pl.labels.cpu().numpy().copy(), num_fg=int(pl.num_fg.item()), losses={k: float(v) for k, v in out.items()},
               logits=pl.logits.float().cpu().view(-1, cfg.DATA.NUM_CLASSES).clone())
    model.backward()
    torch.cuda.synchronize()
    res["grads"] = {k: v.clone() for k, v in model.reference_grads().items()}
    res["acts"] = model.debug_activations()
    del model
    torch.cuda.empty_cache()
    return dict(cfg=cfg, params=params, batch=batch, names=names, ref=ref, aux=aux, ref_grads=ref_grads, bf16=res)


def test_retinanet_r101_bf16_full_size_matches_oracle(r101):
    from basedet_amd.models import params as P
    from oracle.model import Oracle
    cfg, names, ref, aux, got = r101["cfg"], r101["names"], r101["ref"], r101["aux"], r101["bf16"]
    assert got["labels"].shape == (N, 201600)
    assert np.array_equal(got["labels"], aux["labels"])
    assert got["num_fg"] == aux["num_fg"]
    for k in ("cls_loss", "reg_loss", "total_loss"):
        assert abs(got["losses"][k] - ref[k]) / abs(ref[k]) < 2e-2, (k, got["losses"][k], ref[k])
    rl = _rel(got["logits"], aux["logits"])
    print("R101 logits rel-L2 vs the fp32 oracle:", rl)
    assert rl < 2e-2, rl
    orc2 = Oracle(r101["params"], P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=got["acts"])
    l2, _ = orc2.retinanet_losses(r101["batch"])
    g2 = orc2.grads(l2["total_loss"])
    worst_p, worst_i = ("", 0.0), ("", 0.0)
    for n in names:
        rp, ri = _rel(got["grads"][n], r101["ref_grads"][n]), _rel(got["grads"][n], g2[n].detach())
        worst_p = (n, rp) if rp > worst_p[1] else worst_p
        worst_i = (n, ri) if ri > worst_i[1] else worst_i
    print(f"[RetinaNet-R101 2x800x1344] worst per-parameter gradient rel-L2: plain oracle {worst_p}, injected oracle {worst_i}")
    assert worst_i[1] < 2e-2, worst_i
    assert worst_p[1] < 0.5, worst_p          # bf16 forward differences flip ReLU gates near zero (33 blocks): loose by construction
    from tests.test_fullsize_parity_gpu import _check_forward_layers
    _check_forward_layers(Oracle(r101["params"], P.oracle_arch(cfg), record={"_compare": got["acts"]}), r101["batch"], "retinanet_losses",
                          "RetinaNet-R101 2x800x1344", bound=3e-2)


def _fp8_run(r101, dgrad, wgrad):
    """One fp8 forward + backward of the R101 fixture's batch (cached in the fixture: the (True, 2) case is judged against (True, 0) and
    computes it itself when it has not run -- under -k, xdist or any other order)."""
    cache = r101.setdefault("_fp8", {})
    if (dgrad, wgrad) in cache:
        return cache[(dgrad, wgrad)]
    from basedet_amd.models import RetinaNet
    cfg, names, b16 = r101["cfg"], r101["names"], r101["bf16"]
    cfg.MODEL.WEIGHT_DTYPE = "fp8_e4m3"
    cfg.MODEL.FP8_DGRAD = dgrad
    cfg.MODEL.FP8_WGRAD = wgrad
    cfg.MODEL.FP8_STOCHASTIC_ROUNDING = False
    try:
        m8 = RetinaNet(cfg, params=r101["params"])
    finally:
        for k in ("WEIGHT_DTYPE", "FP8_DGRAD", "FP8_WGRAD", "FP8_STOCHASTIC_ROUNDING"):
            cfg.MODEL.pop(k, None)
    assert any(c.fp8 for c in m8.convs.values())
    assert any(c.fp8_dgrad for c in m8.convs.values()) == dgrad
    assert any(c.fp8_wgrad for c in m8.convs.values()) == (wgrad > 0)
    out8 = m8(r101["batch"])
    m8.backward()
    torch.cuda.synchronize()
    g8 = m8.reference_grads()
    a = torch.cat([g8[n].double().reshape(-1) for n in names])
    b = torch.cat([b16["grads"][n].double().reshape(-1) for n in names])
    worst = min((float(torch.dot(g8[n].double().reshape(-1), b16["grads"][n].double().reshape(-1)) /
                       (g8[n].double().norm() * b16["grads"][n].double().norm())), n)
                for n in names if n.endswith(".weight") and g8[n].dim() == 4 and g8[n].shape[-1] == 3)
    res = dict(losses={k: float(v) for k, v in out8.items()}, labels=m8._cur.labels.cpu().numpy().copy(), finite=bool(torch.isfinite(a).all()),
               cos=float(torch.dot(a, b) / (a.norm() * b.norm())), worst=worst)
    del m8
    torch.cuda.empty_cache()
    cache[(dgrad, wgrad)] = res
    return res


@pytest.mark.parametrize("dgrad,wgrad", [(False, 0), (True, 0), (True, 2)])
def test_retinanet_r101_fp8_full_size_tolerance(r101, dgrad, wgrad):
    """WEIGHT_DTYPE = fp8_e4m3 on R101: forward only (dgrad False), with e5m2 data gradients under per-group delayed scales, and with the
    3x3 weight gradients from the one-byte twins as well (FP8_WGRAD = 2: conv_wgrad3x3_fp8_kernel, the default since round 4)."""
    ref, aux, b16 = r101["ref"], r101["aux"], r101["bf16"]
    got = _fp8_run(r101, dgrad, wgrad)
    assert np.array_equal(got["labels"], aux["labels"])
    for k in ("cls_loss", "reg_loss", "total_loss"):
        v8, v16, vr = got["losses"][k], b16["losses"][k], ref[k]
        print(f"{k}: fp8 {v8:.5f} bf16 {v16:.5f} fp32 oracle {vr:.5f}")
        assert abs(v8 - vr) / abs(vr) < 5e-2, (k, v8, vr)
    assert got["finite"]
    cos, worst = got["cos"], got["worst"]
    print(f"R101 gradient cosine fp8 (dgrad={dgrad}, wgrad={wgrad}) vs bf16: {cos:.5f}")
    # floors with real margin (observed 0.9814-0.9818 with e5m2 data gradients), and the one-byte weight gradients are
    # judged AGAINST the same run without them (computed here if that case has not run): what they may cost is a delta
    assert cos >= 0.975, cos
    # ... and layer by layer over the 3x3 weights (what the one-byte weight-gradient kernel produces when wgrad is on): a two-image batch at
    # the pre-probe scales; the lowest layer is a thin backbone conv2 or, with fp8 weight gradients, the deepest conv of the classification tower
    print(f"worst 3x3 weight-gradient cosine (dgrad={dgrad}, wgrad={wgrad}):", worst)
    assert worst[0] >= 0.84, worst            # observed: 0.910 forward only (layer2.0.conv2), 0.895 with e5m2 data gradients, 0.869 with fp8 weight gradients
    if wgrad:
        base = _fp8_run(r101, dgrad, 0)
        assert cos >= base["cos"] - 0.003, (cos, base["cos"])          # observed: -0.0004
        assert worst[0] >= base["worst"][0] - 0.05, (worst, base["worst"])      # observed: -0.026
