import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_model_gpu import _setup
from basedet_amd.models import RetinaNet, params as P
from oracle.model import Oracle
for backbone, N, size in (("resnet18", 2, (128, 160)), ("resnet50", 3, (96, 128))):
    cfg, params, batch = _setup(backbone, N, size)
    model = RetinaNet(cfg, params=params)
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    losses = model(batch)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=model.debug_activations())
    ref_losses, aux = orc.retinanet_losses(batch)
    ref_grads = orc.grads(ref_losses["total_loss"])
    model.backward()
    torch.cuda.synchronize()
    print(backbone, {k: (float(v), float(ref_losses[k])) for k, v in losses.items()})
    ent = [e[0] for e in model.arena.entries]
    for name in names:
        idx = ent.index(name)
        g = model.arena.view("g", idx).detach().cpu()
        r = ref_grads[name].detach()
        if g.ndim == 4: g = g.permute(0, 3, 1, 2)
        g = g[: r.shape[0]].double().reshape(-1); r = r.double().reshape(-1)
        rel = float((g - r).norm() / (r.norm() + 1e-30)); cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-30))
        flag = "  <<<" if rel > 0.05 else ""
        print(f"{name:55s} rel={rel:.4f} cos={cos:.5f} |r|={float(r.norm()):.3e}{flag}")
