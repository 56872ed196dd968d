"""GPU parity of the FreeAnchor bag losses (bd_freeanchor_loss_fwd_bwd, models/det/free_anchor.py:38-142) against the torch-CPU
oracle (oracle/freeanchor.py): loss values to 1e-4 relative (fp32 sums in a different order), gradients to bf16 storage precision.
Sizes: a 3-level pyramid with 9 anchors per location, up to 6 ground-truth boxes per image."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(seed, N=2, K=16, sizes=((16, 20), (8, 10), (4, 5)), strides=(8, 16, 32), gmax=6, dup_class=False):
    from oracle import box_ops
    rng = np.random.default_rng(seed)
    scales = [[s * 4, s * 4 * 2 ** (1 / 3), s * 4 * 2 ** (2 / 3)] for s in strides]
    anchors = np.concatenate(box_ops.default_anchors(list(sizes), list(strides), scales, [[0.5, 1, 2]], 0.5), 0).astype(np.float32)
    A = anchors.shape[0]
    H, W = sizes[0][0] * strides[0], sizes[0][1] * strides[0]
    gt = np.zeros((N, gmax, 5), np.float32)
    num = np.zeros((N,), np.int32)
    for n in range(N):
        g = int(rng.integers(1, gmax + 1)) if n > 0 else gmax
        num[n] = g
        cx, cy = rng.uniform(20, W - 20, g), rng.uniform(20, H - 20, g)
        w, h = rng.uniform(16, 90, g), rng.uniform(16, 90, g)
        gt[n, :g, 0] = np.clip(cx - w / 2, 0, W); gt[n, :g, 1] = np.clip(cy - h / 2, 0, H)
        gt[n, :g, 2] = np.clip(cx + w / 2, 0, W); gt[n, :g, 3] = np.clip(cy + h / 2, 0, H)
        gt[n, :g, 4] = rng.integers(1, K + 1, g)
        if dup_class and g >= 2:            # two overlapping boxes of one class: the later one must win shared anchors
            gt[n, 1, :4] = gt[n, 0, :4] + np.array([2, 2, 2, 2], np.float32)
            gt[n, 1, 4] = gt[n, 0, 4]
    # predictions that overlap the gts well for some anchors: offsets = encoded gt of the best-matching gt + noise
    offsets = np.zeros((N, A, 4), np.float32)
    for n in range(N):
        best = box_ops.box_iou(gt[n, :max(1, num[n]), :4], anchors).argmax(0)
        tgt = box_ops.box_encode(anchors, gt[n, best, :4], (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2))
        offsets[n] = np.clip(tgt, -8, 8) * rng.uniform(0.6, 1.0, (A, 1)).astype(np.float32) + rng.normal(0, 0.3, (A, 4)).astype(np.float32)
    logits = rng.normal(-2.0, 1.5, (N, A, K)).astype(np.float32)
    return anchors, gt, num, logits, offsets


@pytest.mark.parametrize("seed,dup,beta", [(0, False, 0.0), (1, True, 0.0), (2, False, 0.11)])
def test_freeanchor_losses_and_gradients(seed, dup, beta):
    from basedet_amd import ops
    from oracle import freeanchor
    anchors, gt, num, logits, offsets = _problem(seed, dup_class=dup)
    N, A, K = logits.shape
    apix, ld = 9, 40
    bucket = 50 if seed != 2 else 20
    lg = torch.from_numpy(logits).to(torch.bfloat16)
    of = torch.from_numpy(offsets).to(torch.bfloat16)
    # ---- oracle on the bf16-rounded operands
    lt = lg.float().clone().requires_grad_(True)
    ot = of.float().clone().requires_grad_(True)
    pos, neg = freeanchor.bag_losses(lt, ot, anchors, gt, num, std=(0.1, 0.1, 0.2, 0.2), iou_thresh=0.6, bucket=bucket, beta=beta,
                                     reg_weight=0.75, alpha=0.25, gamma=2.0)
    (pos + neg).backward()
    # ---- HIP
    dev = "cuda"
    off_dev = torch.zeros((N * (A // apix), ld), dtype=torch.bfloat16, device=dev)
    off_dev[:, :apix * 4] = of.reshape(N * (A // apix), apix * 4).to(dev)
    lg_dev = lg.reshape(N * A, K).to(dev)
    d_lg = torch.full_like(lg_dev, 7.0)
    d_of = torch.full_like(off_dev, 7.0)
    loss = torch.zeros(2, dtype=torch.float32, device=dev)
    ws = torch.empty(ops.freeanchor_workspace_bytes(N, gt.shape[1], bucket, A), dtype=torch.uint8, device=dev)
    ops.freeanchor_loss_fwd_bwd(lg_dev, off_dev, ld, apix, torch.from_numpy(anchors).to(dev), K, torch.from_numpy(gt).to(dev),
                                torch.from_numpy(num).to(dev), (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), 0.6, bucket, beta, 0.75, 0.25, 2.0,
                                loss, d_lg, d_of, ws)
    torch.cuda.synchronize()
    got = loss.cpu().numpy()
    assert abs(got[0] - float(pos)) <= 1e-4 * abs(float(pos)) + 1e-7, (got[0], float(pos))
    assert abs(got[1] - float(neg)) <= 1e-4 * abs(float(neg)) + 1e-7, (got[1], float(neg))
    # gradients: bf16 storage (8 bits of mantissa) on top of fp32 arithmetic in another order
    g_l = d_lg.float().cpu().reshape(N, A, K)
    r_l = lt.grad
    assert float((g_l - r_l).norm() / r_l.norm()) < 6e-3
    assert float((g_l - r_l).abs().max()) <= 1e-2 * float(r_l.abs().max()) + 1e-9
    g_o = d_of.float().cpu()[:, :apix * 4].reshape(N, A, 4)
    r_o = ot.grad
    assert float(r_o.abs().max()) > 0
    assert float((g_o - r_o).norm() / r_o.norm()) < 6e-3
    assert (d_of.float().cpu()[:, apix * 4:] == 0).all()                  # padding columns of bbox_pred stay zero
    # anchors outside every bag get exactly zero offset gradient
    assert ((r_o.abs().sum(-1) == 0) == (g_o.abs().sum(-1) == 0)).all()


def test_freeanchor_image_without_boxes():
    from basedet_amd import ops
    from oracle import freeanchor
    anchors, gt, num, logits, offsets = _problem(3)
    num[1] = 0
    N, A, K = logits.shape
    lt = torch.from_numpy(logits).to(torch.bfloat16).float().requires_grad_(True)
    ot = torch.from_numpy(offsets).to(torch.bfloat16).float().requires_grad_(True)
    pos, neg = freeanchor.bag_losses(lt, ot, anchors, gt, num, bucket=50)
    dev = "cuda"
    off_dev = torch.zeros((N * (A // 9), 40), dtype=torch.bfloat16, device=dev)
    off_dev[:, :36] = ot.detach().to(torch.bfloat16).reshape(-1, 36).to(dev)
    lg_dev = lt.detach().to(torch.bfloat16).reshape(N * A, K).to(dev)
    d_lg, d_of = torch.empty_like(lg_dev), torch.empty_like(off_dev)
    loss = torch.zeros(2, dtype=torch.float32, device=dev)
    ws = torch.empty(ops.freeanchor_workspace_bytes(N, gt.shape[1], 50, A), dtype=torch.uint8, device=dev)
    ops.freeanchor_loss_fwd_bwd(lg_dev, off_dev, 40, 9, torch.from_numpy(anchors).to(dev), K, torch.from_numpy(gt).to(dev),
                                torch.from_numpy(num).to(dev), (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), 0.6, 50, 0.0, 0.75, 0.25, 2.0,
                                loss, d_lg, d_of, ws)
    got = loss.cpu().numpy()
    assert abs(got[0] - float(pos)) <= 1e-4 * abs(float(pos)) and abs(got[1] - float(neg)) <= 1e-4 * abs(float(neg))
    assert (d_of.float().cpu().reshape(N, -1)[1] == 0).all()


def test_freeanchor_training_step_matches_oracle():
    """FreeAnchor(RetinaNet) end to end: losses against the fp32 oracle, then gradients of every trainable parameter against the
    oracle evaluated on the stored bf16 activations."""
    from basedet_amd.configs import FreeAnchorConfig
    from basedet_amd.models import FreeAnchor, params as P
    from basedet_amd.utils import DummyLoader
    from oracle.model import Oracle
    N, size = 2, (128, 160)
    cfg = FreeAnchorConfig()
    cfg.merge(dict(MODEL=dict(BACKBONE=dict(NAME="resnet18", OUT_FEATURE_CHANNELS=[128, 256, 512]), FPN=dict(TOP_BLOCK_IN_CHANNELS=512))))
    cfg.MODEL.BATCHSIZE = N
    params = P.init_retinanet_params(cfg, seed=0)
    batch = next(DummyLoader(N, size, seed=0))
    batch["data"] = (batch["data"] * 255).astype(np.float32)
    model = FreeAnchor(cfg, params=params)
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    ref, _ = orc.freeanchor_losses(batch)
    out = model(batch)
    for k in ("pos_loss", "neg_loss", "total_loss"):
        got, want = float(out[k]), float(ref[k].detach())
        assert abs(got - want) / abs(want) < 2e-2, (k, got, want)
    model.backward()
    torch.cuda.synchronize()
    orc2 = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=model.debug_activations())
    l2, _ = orc2.freeanchor_losses(batch)
    g2 = orc2.grads(l2["total_loss"])
    got = model.reference_grads()
    for n in names:
        r = g2[n].detach().double().reshape(-1)
        g = got[n].double().reshape(-1)
        rel = float((g - r).norm() / (r.norm() + 1e-30))
        assert rel < 2e-2, (n, rel)
