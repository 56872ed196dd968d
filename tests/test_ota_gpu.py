"""GPU parity of the OTA dynamic top-k assignment (bd_ota_assign; models/det/ota.py:76-181, layers/common/matcher.py:123-161)
against the numpy oracle (oracle/box_ops.py ota_ground_truth, which sums the class cost literally over the K one-hot columns).

The costs contain exp/log (hardware intrinsics on the device) and the kernel re-associates the class-cost sum, so the two cost
matrices agree to ~1e-5 relative, not bitwise: a point may be assigned differently only where the costs that decide it are that
close.  The test therefore demands identical labels / targets everywhere except on such near-ties (and at most 0.5 % of the
foreground), and checks the near-tie claim for every differing point."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(seed, N=2, K=16, sizes=((16, 20), (8, 10), (4, 5)), strides=(8, 16, 32), gmax=7):
    from oracle import box_ops
    rng = np.random.default_rng(seed)
    pts = box_ops.point_anchors(list(sizes), list(strides), 0.5, 1)
    allp = np.concatenate(pts, 0).astype(np.float32)
    P = allp.shape[0]
    H, W = sizes[0][0] * strides[0], sizes[0][1] * strides[0]
    gt = np.zeros((N, gmax, 5), np.float32)
    num = np.zeros((N,), np.int32)
    for n in range(N):
        g = gmax if n == 0 else int(rng.integers(1, gmax))
        num[n] = g
        cx, cy = rng.uniform(20, W - 20, g), rng.uniform(20, H - 20, g)
        w, h = rng.uniform(24, 110, g), rng.uniform(24, 110, g)
        gt[n, :g, 0] = np.clip(cx - w / 2, 0, W); gt[n, :g, 1] = np.clip(cy - h / 2, 0, H)
        gt[n, :g, 2] = np.clip(cx + w / 2, 0, W); gt[n, :g, 3] = np.clip(cy + h / 2, 0, H)
        gt[n, :g, 4] = rng.integers(1, K + 1, g)
    # predictions: ltrb towards a random gt with noise (so IoUs are substantial), logits around the prior
    pred = np.zeros((N, P, 4), np.float32)
    for n in range(N):
        j = rng.integers(0, num[n], P)
        b = gt[n, j, :4]
        d = np.stack([allp[:, 0] - b[:, 0], allp[:, 1] - b[:, 1], b[:, 2] - allp[:, 0], b[:, 3] - allp[:, 1]], 1)
        pred[n] = np.maximum(d * rng.uniform(0.7, 1.3, (P, 4)) + rng.normal(0, 2.0, (P, 4)), 0).astype(np.float32)
    logits = rng.normal(-2.5, 1.2, (N, P, K)).astype(np.float32)
    lvl_start = [0]
    for (h, w) in sizes:
        lvl_start.append(lvl_start[-1] + h * w)
    return pts, allp, lvl_start, list(strides), gt, num, logits, pred


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ota_assignment_matches_oracle(seed):
    from basedet_amd import ops
    from oracle import box_ops
    pts, allp, lvl_start, strides, gt, num, logits, pred = _problem(seed)
    N, P, K = logits.shape
    lg = torch.from_numpy(logits).to(torch.bfloat16)
    pr = torch.from_numpy(pred).to(torch.bfloat16)
    lab_o, tgt_o, iou_o, aux = box_ops.ota_ground_truth(pts, strides, lg.float().numpy(), pr.float().numpy(), gt, num, 0.25, 2.0, 1.5, 2.5, 10)
    dev = "cuda"
    labels = torch.full((N, P), -7, dtype=torch.int32, device=dev)
    targets = torch.full((N, P, 4), -7.0, dtype=torch.float32, device=dev)
    ious = torch.full((N, P), -7.0, dtype=torch.float32, device=dev)
    stats = torch.zeros(2, dtype=torch.float32, device=dev)
    ws = torch.empty(ops.ota_assign_workspace_bytes(N, P), dtype=torch.uint8, device=dev)
    ops.ota_assign(torch.from_numpy(allp).to(dev), lvl_start, strides, lg.reshape(N * P, K).to(dev), K, pr.reshape(N * P, 4).to(dev),
                   torch.from_numpy(gt).to(dev), torch.from_numpy(num).to(dev), 0.25, 2.0, 1.5, 2.5, 10, labels, targets, ious, stats, ws)
    torch.cuda.synchronize()
    lab, tgt, iou_t = labels.cpu().numpy(), targets.cpu().numpy(), ious.cpu().numpy()
    assert (lab_o > 0).sum() >= 10
    diff = np.argwhere(lab != lab_o)
    assert len(diff) <= max(1, int(0.005 * (lab_o > 0).sum())), (len(diff), int((lab_o > 0).sum()))
    for n, p in diff:                       # every difference sits on a near-tie of the deciding costs
        cost = aux[n][0][:, p]
        near = []
        for g in range(cost.shape[0]):
            kth = np.sort(aux[n][0][g])[:12]
            near.append(np.min(np.abs(kth - cost[g]) / np.maximum(np.abs(cost[g]), 1e-6)))
        two = np.sort(cost)[:2]
        assert min(near) < 1e-4 or abs(two[1] - two[0]) / max(abs(two[0]), 1e-6) < 1e-4, (n, p, cost)
    same = lab == lab_o
    fg = same & (lab_o > 0)
    # where the assignment agrees the targets are the same numbers (ltrb exact; IoU to fp32 rounding)
    np.testing.assert_array_equal(tgt[fg], tgt_o[fg])
    np.testing.assert_allclose(iou_t[fg], iou_o[fg], rtol=2e-6, atol=1e-7)
    assert (tgt[lab == 0] == 0).all() and (iou_t[lab == 0] == 0).all()
    st = stats.cpu().numpy()
    assert st[0] == (lab > 0).sum() and st[1] == 2 * st[0]


def test_ota_training_step_matches_oracle():
    """OTA(FCOS) end to end: the assignment against the oracle (near-tie tolerance as above), the three losses against the fp32
    oracle, gradients against the oracle on the stored bf16 activations with the device's own targets."""
    from basedet_amd.configs import OTAConfig
    from basedet_amd.models import OTA, params as P
    from basedet_amd.utils import DummyLoader
    from oracle.model import Oracle
    N, size = 2, (128, 160)
    cfg = OTAConfig()
    cfg.MODEL.BATCHSIZE = N
    params = P.init_fcos_params(cfg, seed=0, residual_gamma=0.25)
    params["head.bbox_pred.bias"] = np.full_like(params["head.bbox_pred.bias"], 0.5)
    batch = next(DummyLoader(N, size, seed=0))
    batch["data"] = (batch["data"] * 255).astype(np.float32)
    model = OTA(cfg, params=params)
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    ref, aux = orc.ota_losses(batch)
    out = model(batch)
    pl = model._cur
    lab = pl.labels.cpu().numpy()
    nfg = int((aux["labels"] > 0).sum())
    assert nfg >= 5
    # the oracle's network is fp32, the device's bf16: the cost matrices differ at the 1e-2 level, so compare the assignment as a set
    agree = (lab == aux["labels"]).mean()
    assert agree > 0.995, agree
    assert abs(int((lab > 0).sum()) - nfg) <= max(2, nfg // 5)
    for k in ("loss_cls", "loss_offsets", "loss_ious", "total_loss"):
        got, want = float(out[k]), float(ref[k].detach())
        assert abs(got - want) / abs(want) < 0.1, (k, got, want)
    model.backward()
    torch.cuda.synchronize()
    forced = (lab, pl.gt_offsets.cpu().numpy(), pl.gt_ctr.cpu().numpy())
    orc2 = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=model.debug_activations())
    l2, _ = orc2.ota_losses(batch, forced=forced)
    for k in ("loss_cls", "loss_offsets", "loss_ious", "total_loss"):       # same activations, same targets: tight
        got, want = float(out[k]), float(l2[k].detach())
        assert abs(got - want) / abs(want) < 2e-2, (k, got, want)
    g2 = orc2.grads(l2["total_loss"])
    got = model.reference_grads()
    for n in names:
        r = g2[n].detach().double().reshape(-1)
        g = got[n].double().reshape(-1)
        rel = float((g - r).norm() / (r.norm() + 1e-30))
        assert rel < 2e-2, (n, rel)


@pytest.mark.parametrize("seed", [0, 3])
def test_ota_sinkhorn_assignment_matches_oracle(seed):
    """MATCHING = "sinkhorn" (layers/common/matcher.py:106-121): 50 log-domain Sinkhorn updates in fp32 on both sides; the plans agree
    to rounding, the assignment except where the two largest rescaled plan entries of a point are within 1e-3 of each other."""
    from basedet_amd import ops
    from oracle import box_ops
    pts, allp, lvl_start, strides, gt, num, logits, pred = _problem(seed, gmax=5)
    N, P, K = logits.shape
    lg = torch.from_numpy(logits).to(torch.bfloat16)
    pr = torch.from_numpy(pred).to(torch.bfloat16)
    lab_o, tgt_o, iou_o, aux = box_ops.ota_ground_truth(pts, strides, lg.float().numpy(), pr.float().numpy(), gt, num, 0.25, 2.0, 1.5, 2.5,
                                                        10, matching="sinkhorn")
    dev = "cuda"
    labels = torch.full((N, P), -7, dtype=torch.int32, device=dev)
    targets = torch.full((N, P, 4), -7.0, dtype=torch.float32, device=dev)
    ious = torch.full((N, P), -7.0, dtype=torch.float32, device=dev)
    stats = torch.zeros(2, dtype=torch.float32, device=dev)
    ws = torch.empty(ops.ota_sinkhorn_workspace_bytes(N, P, gt.shape[1]), dtype=torch.uint8, device=dev)
    ops.ota_assign_sinkhorn(torch.from_numpy(allp).to(dev), lvl_start, strides, lg.reshape(N * P, K).to(dev), K, pr.reshape(N * P, 4).to(dev),
                            torch.from_numpy(gt).to(dev), torch.from_numpy(num).to(dev), 0.25, 2.0, 1.5, 2.5, labels, targets, ious, stats, ws)
    torch.cuda.synchronize()
    lab, tgt, iou_t = labels.cpu().numpy(), targets.cpu().numpy(), ious.cpu().numpy()
    nfg = int((lab_o > 0).sum())
    assert nfg >= 5
    diff = np.argwhere(lab != lab_o)
    assert len(diff) <= max(1, nfg // 20), (len(diff), nfg)
    for n, p in diff:
        col = np.sort(aux[n][1][:, p])[::-1]
        assert col[0] - col[1] < 1e-3 * max(col[0], 1e-30) or not np.isfinite(col[0]), (n, p, col[:3])
    fg = (lab == lab_o) & (lab_o > 0)
    np.testing.assert_array_equal(tgt[fg], tgt_o[fg])
    np.testing.assert_allclose(iou_t[fg], iou_o[fg], rtol=2e-6, atol=1e-7)
    st = stats.cpu().numpy()
    assert st[0] == (lab > 0).sum() and st[1] == 2 * st[0]


def test_ota_sinkhorn_model_step():
    """OTA with MATCHING = "sinkhorn" end to end: finite losses of the expected structure and a backward pass."""
    from basedet_amd.configs import OTAConfig
    from basedet_amd.models import OTA, params as P
    from basedet_amd.utils import DummyLoader
    N, size = 2, (128, 160)
    cfg = OTAConfig()
    cfg.MODEL.MATCHING = "sinkhorn"
    cfg.MODEL.BATCHSIZE = N
    params = P.init_fcos_params(cfg, seed=0, residual_gamma=0.25)
    params["head.bbox_pred.bias"] = np.full_like(params["head.bbox_pred.bias"], 0.5)
    batch = next(DummyLoader(N, size, seed=0))
    batch["data"] = (batch["data"] * 255).astype(np.float32)
    model = OTA(cfg, params=params)
    out = model(batch)
    vals = {k: float(v) for k, v in out.items()}
    assert all(np.isfinite(v) for v in vals.values()) and vals["loss_cls"] > 0 and vals["loss_offsets"] > 0
    assert abs(vals["total_loss"] - (vals["loss_cls"] + vals["loss_offsets"] + vals["loss_ious"])) < 1e-4 * vals["total_loss"]
    nfg = float(model._cur.stats[0])
    assert nfg >= 1 and nfg == float((model._cur.labels > 0).sum())
    model.backward()
    torch.cuda.synchronize()
    g = model.reference_grads()
    assert all(torch.isfinite(t).all() for t in g.values())
