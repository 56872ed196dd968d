"""GPU parity of the box operators / losses against the numpy oracle (oracle/box_ops.py) and the reference's
own known-answer vectors (tests/golden/reference_kat.npz).  Integer outputs and IoU-derived decisions are
compared bit-exactly; log/exp based values with a stated tolerance."""
import os

import numpy as np
import pytest
import torch

from oracle import box_ops as ob

pytestmark = pytest.mark.gpu

SCALES = [[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]]
RATIOS = [[0.5, 1, 2]]
STRIDES = [8, 16, 32, 64, 128]


def _ops():
    from basedet_amd import ops
    return ops


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _anchors_dev(sizes):
    ops = _ops()
    outs = []
    sc = np.asarray(SCALES, np.float32).tolist()
    ra = np.asarray(RATIOS, np.float32).tolist() * 5
    for (h, w), s, scl, rat in zip(sizes, STRIDES, sc, ra):
        base = _dev(ob.generate_base_anchors(scl, rat))
        out = torch.empty((h * w * base.shape[0], 4), dtype=torch.float32, device="cuda")
        ops.anchors_generate(h, w, s, 0.5, base, out)
        outs.append(out)
    return outs


def test_anchors_bit_exact():
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    ref = ob.default_anchors(sizes, STRIDES, SCALES, RATIOS, 0.5)
    got = _anchors_dev(sizes)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g.cpu().numpy())
    ops = _ops()
    pts = ob.point_anchors(sizes, STRIDES, 0.5, 1)
    for (h, w), s, r in zip(sizes, STRIDES, pts):
        out = torch.empty((h * w, 2), dtype=torch.float32, device="cuda")
        ops.points_generate(h, w, s, 0.5, 1, out)
        assert np.array_equal(r, out.cpu().numpy())


def test_reference_known_answers(golden_dir):
    """tests/structures/test_boxes.py:38-86 and tests/layers/test_postprocess.py:13-28 of the reference."""
    ops = _ops()
    k = np.load(os.path.join(golden_dir, "reference_kat.npz"))
    b1, b2 = _dev(k["boxes1"]), _dev(k["boxes2"])
    assert np.allclose(ops.box_pairwise(b1, b2, 0).cpu().numpy(), k["iou_1x2"])
    assert np.allclose(ops.box_pairwise(b2, b1, 1).cpu().numpy(), k["ioa_2x1"])
    assert np.allclose(ops.box_pairwise(b1, b2, 2).cpu().numpy(), k["inter_1x2"])
    keep = ops.batched_nms(_dev(k["nms_boxes"]), _dev(k["nms_scores"]), _dev(k["nms_labels"]), float(k["nms_iou_thresh"]))
    assert keep.cpu().numpy().tolist() == k["nms_keep"].tolist()


def test_pairwise_bit_exact_random():
    ops = _ops()
    rng = np.random.default_rng(0)
    def boxes(n):
        xy = rng.uniform(0, 800, (n, 2)).astype(np.float32)
        wh = rng.uniform(1, 300, (n, 2)).astype(np.float32)
        return np.concatenate([xy, xy + wh], 1).astype(np.float32)
    a, b = boxes(37), boxes(1001)
    for mode, fn in ((0, ob.box_iou), (1, ob.box_ioa), (2, ob.intersection), (3, ob.box_giou)):
        got = ops.box_pairwise(_dev(a), _dev(b), mode).cpu().numpy()
        assert np.array_equal(got, fn(a, b)), f"mode {mode}"
    # empty inputs
    assert ops.box_pairwise(_dev(a[:0]), _dev(b), 0).shape == (0, 1001)


def test_box_coder():
    ops = _ops()
    rng = np.random.default_rng(1)
    xy = rng.uniform(0, 500, (5000, 2)).astype(np.float32); wh = rng.uniform(4, 200, (5000, 2)).astype(np.float32)
    anc = np.concatenate([xy, xy + wh], 1)
    xy = rng.uniform(0, 500, (5000, 2)).astype(np.float32); wh = rng.uniform(4, 200, (5000, 2)).astype(np.float32)
    gt = np.concatenate([xy, xy + wh], 1)
    mean, std = (0.0, 0.0, 0.0, 0.0), (0.1, 0.1, 0.2, 0.2)
    enc = ops.box_encode(_dev(anc), _dev(gt), mean, std).cpu().numpy()
    ref = ob.box_encode(anc, gt, mean, std)
    assert np.array_equal(enc[:, :2], ref[:, :2])                        # no transcendental: bit-exact
    assert np.allclose(enc[:, 2:], ref[:, 2:], rtol=2e-6, atol=2e-6)      # logf: ulp-level tolerance
    dec = ops.box_decode(_dev(anc), _dev(ref), mean, std).cpu().numpy()
    assert np.allclose(dec, ob.box_decode(anc, ref, mean, std), rtol=1e-5, atol=1e-3)
    assert np.allclose(dec, gt, rtol=1e-4, atol=1e-2)                     # encode -> decode round trip


@pytest.mark.parametrize("size", [(800, 1344), (512, 512)])
def test_retina_assign_matches_oracle(golden_dir, size):
    """RetinaNet.get_ground_truth on the DummyLoader annotation pattern: labels / matched indices bit-exact."""
    ops = _ops()
    tag = f"{size[0]}x{size[1]}"
    d = np.load(os.path.join(golden_dir, "dummy_loader.npz"))
    anno, info = d[f"anno_{tag}"], d[f"im_info_{tag}"]
    N = 5
    gt = ob.tile_batch(anno, N)
    ng = ob.tile_batch(info, N)[:, 4].astype(np.int32)
    ng[-1] = 0                                        # an image without boxes
    sizes = [((size[0] + s - 1) // s, (size[1] + s - 1) // s) for s in STRIDES]
    anchors = torch.cat(_anchors_dev(sizes))
    A = anchors.shape[0]
    labels = torch.empty((N, A), dtype=torch.int32, device="cuda")
    midx = torch.empty((N, A), dtype=torch.int32, device="cuda")
    offs = torch.empty((N, A, 4), dtype=torch.float32, device="cuda")
    nfg = torch.zeros((1,), dtype=torch.int32, device="cuda")
    ws = torch.empty((N * gt.shape[1],), dtype=torch.float32, device="cuda")
    ops.retina_assign_encode(anchors, _dev(gt), _dev(ng), 0.4, 0.5, True, (0, 0, 0, 0), (1, 1, 1, 1), labels, midx, offs, nfg, ws)
    rl, ro, ri = ob.retinanet_ground_truth(anchors.cpu().numpy(), gt, ng)
    assert np.array_equal(labels.cpu().numpy(), rl)
    assert np.array_equal(midx.cpu().numpy(), ri)
    assert int(nfg.item()) == int((rl > 0).sum())
    go = offs.cpu().numpy()
    assert np.array_equal(go[..., :2], ro[..., :2])
    assert np.allclose(go[..., 2:], ro[..., 2:], rtol=2e-6, atol=2e-6)
    assert (rl > 0).sum() > 0 and (rl == -1).sum() > 0


def test_fcos_assign_matches_oracle(golden_dir):
    ops = _ops()
    d = np.load(os.path.join(golden_dir, "dummy_loader.npz"))
    N = 3
    gt = ob.tile_batch(d["anno_800x1344"], N)
    ng = ob.tile_batch(d["im_info_800x1344"], N)[:, 4].astype(np.int32)
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    pts = ob.point_anchors(sizes, STRIDES, 0.5, 1)
    soi = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, float("inf")]]
    rl, ro, rc = ob.fcos_ground_truth(pts, STRIDES, gt, ng, soi, 1.5)
    P = sum(p.shape[0] for p in pts)
    start = np.cumsum([0] + [p.shape[0] for p in pts]).tolist()
    labels = torch.empty((N, P), dtype=torch.int32, device="cuda")
    offs = torch.empty((N, P, 4), dtype=torch.float32, device="cuda")
    ctr = torch.empty((N, P), dtype=torch.float32, device="cuda")
    stats = torch.zeros((2,), dtype=torch.float32, device="cuda")
    ops.fcos_assign(_dev(np.concatenate(pts)), start, soi, STRIDES, 1.5, _dev(gt), _dev(ng), labels, offs, ctr, stats)
    assert np.array_equal(labels.cpu().numpy(), rl)
    assert np.array_equal(offs.cpu().numpy(), ro)
    fg = rl > 0
    assert np.array_equal(ctr.cpu().numpy()[fg], rc[fg])
    st = stats.cpu().numpy()
    assert st[0] == fg.sum() and np.isclose(st[1], rc[fg].sum(), rtol=1e-5)


def test_nms_random_matches_oracle():
    ops = _ops()
    rng = np.random.default_rng(3)
    for n in (1, 63, 64, 65, 700, 3000):
        xy = rng.uniform(0, 300, (n, 2)).astype(np.float32); wh = rng.uniform(5, 120, (n, 2)).astype(np.float32)
        boxes = np.concatenate([xy, xy + wh], 1)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        scores[: n // 3] = scores[0]                      # ties: stable order decides
        labels = rng.integers(0, 5, n).astype(np.int32)
        for mo in (None, 100):
            ref = ob.batched_nms(boxes, scores, labels, 0.5, mo)
            got = ops.batched_nms(_dev(boxes), _dev(scores), _dev(labels), 0.5, mo).cpu().numpy()
            assert got.tolist() == ref.tolist(), (n, mo)
    assert ops.batched_nms(_dev(np.zeros((0, 4), np.float32)), _dev(np.zeros((0,), np.float32)), None, 0.5).numel() == 0


def test_nms_matches_reference_py_cpu_nms(golden_dir):
    """The HIP NMS against keep lists generated by the reference's own numpy NMS (tests/golden/reference_nms.npz <- make_golden.py)."""
    ops = _ops()
    k = np.load(os.path.join(golden_dir, "reference_nms.npz"))
    for i in k["cases"]:
        boxes, scores, thr = k[f"boxes_{i}"], k[f"scores_{i}"], float(k[f"thr_{i}"])
        got = ops.batched_nms(_dev(boxes), _dev(scores), None, thr).cpu().numpy()
        assert got.tolist() == k[f"keep_{i}"].tolist(), int(i)
        got = ops.batched_nms(_dev(boxes), _dev(scores), _dev(np.full(len(scores), 3, np.int32)), thr).cpu().numpy()
        assert got.tolist() == k[f"keep_{i}"].tolist(), int(i)


def test_focal_and_l1_losses():
    """sigmoid_focal_loss / smooth_l1_loss value and gradient vs the float64 oracle (tolerance: 2e-3 rel on the
    sum -- bf16 logits are exact inputs to both sides; gradients compared after the kernel's bf16 rounding)."""
    ops = _ops()
    rng = np.random.default_rng(5)
    rows, K = 4000, 80
    x = torch.from_numpy(rng.normal(0, 3, (rows, K)).astype(np.float32)).to(torch.bfloat16)
    labels = rng.integers(-1, 81, rows).astype(np.int32)
    labels[rng.uniform(size=rows) < 0.8] = 0
    nfg = int((labels > 0).sum())
    norm = torch.tensor([nfg], dtype=torch.int32, device="cuda")
    loss = torch.zeros((1,), dtype=torch.float32, device="cuda")
    dl = torch.empty((rows, K), dtype=torch.bfloat16, device="cuda")
    xf = x.float().numpy().astype(np.float64)
    t = np.zeros((rows, K)); fg = labels > 0
    t[fg, labels[fg] - 1] = 1
    valid = labels >= 0
    # the gamma == 2 instance (default), the general kernel on the same inputs, and a non-integer gamma through the general kernel
    for fast, alpha, gamma in ((1, 0.25, 2.0), (0, 0.25, 2.0), (1, 0.25, 1.5), (1, -1.0, 2.0)):
        loss.zero_()
        ops.focal_loss_fwd_bwd(x.cuda(), _dev(labels), rows, K, alpha, gamma, norm, 1.0, loss, dl, general=not fast)
        ref_loss = ob.sigmoid_focal_loss(xf[valid], t[valid], alpha, gamma).sum() / max(1, nfg)
        assert abs(float(loss.item()) - ref_loss) / ref_loss < 2e-3, (fast, alpha, gamma)
        got = dl.float().cpu().numpy()
        if gamma == 2.0 and alpha == 0.25:
            ref_grad = ob.sigmoid_focal_loss_grad(xf, t, alpha, gamma) * valid[:, None] / max(1, nfg)
            assert np.allclose(got, ref_grad, rtol=2e-2, atol=1e-7), fast
        assert np.all(got[~valid] == 0)
    # smooth L1 (beta = 0 -> L1) with padded channel layout: A = 9 anchors, ld = 40
    pixels, A, ld = 500, 9, 40
    pred = torch.from_numpy(rng.normal(0, 1, (pixels, ld)).astype(np.float32)).to(torch.bfloat16)
    tgt = rng.normal(0, 1, (pixels * A, 4)).astype(np.float32)
    lab = rng.integers(-1, 5, pixels * A).astype(np.int32)
    nfg = int((lab > 0).sum())
    norm = torch.tensor([float(nfg)], dtype=torch.float32, device="cuda")
    for beta in (0.0, 0.11):
        loss.zero_()
        dp = torch.full((pixels, ld), 3.0, dtype=torch.bfloat16, device="cuda")
        ops.smooth_l1_fwd_bwd(pred.cuda(), _dev(tgt), _dev(lab), pixels, A, ld, beta, norm, 1.0, loss, dp)
        p = pred.float().numpy()[:, : A * 4].reshape(-1, 4).astype(np.float64)
        ref = ob.smooth_l1_loss(p[lab > 0], tgt[lab > 0], beta).sum() / max(1, nfg)
        assert abs(float(loss.item()) - ref) / ref < 1e-4
        g = dp.float().cpu().numpy()
        assert np.all(g[:, A * 4:] == 0)
        gg = g[:, : A * 4].reshape(-1, 4)
        assert np.all(gg[lab <= 0] == 0)
        d = p - tgt
        refg = np.sign(d) if beta < 1e-5 else np.where(np.abs(d) < beta, d / beta, np.sign(d))
        assert np.allclose(gg[lab > 0], refg[lab > 0] / max(1, nfg), rtol=1e-2, atol=1e-9)


def test_giou_and_bce_losses():
    ops = _ops()
    rng = np.random.default_rng(7)
    rows = 3000
    pred = torch.from_numpy(rng.uniform(0.5, 60, (rows, 4)).astype(np.float32)).to(torch.bfloat16)
    tgt = rng.uniform(0.5, 60, (rows, 4)).astype(np.float32)
    w = rng.uniform(0, 1, rows).astype(np.float32)
    lab = (rng.uniform(size=rows) < 0.5).astype(np.int32)
    norm = torch.tensor([float(w[lab > 0].sum())], dtype=torch.float32, device="cuda")
    loss = torch.zeros((1,), dtype=torch.float32, device="cuda")
    dp = torch.empty((rows, 4), dtype=torch.bfloat16, device="cuda")
    ops.giou_ltrb_fwd_bwd(pred.cuda(), _dev(tgt), _dev(w), _dev(lab), rows, norm, 1.0, loss, dp)
    p = pred.float().numpy().astype(np.float64)
    fg = lab > 0
    ref = (ob.iou_loss_ltrb(p[fg], tgt[fg], "giou") * w[fg]).sum() / max(1.0, float(w[fg].sum()))
    assert abs(float(loss.item()) - ref) / ref < 1e-4
    # numeric gradient of the oracle expression
    eps = 1e-4
    gnum = np.zeros((rows, 4))
    for k in range(4):
        pp = p.copy(); pp[:, k] += eps
        pm = p.copy(); pm[:, k] -= eps
        gnum[:, k] = (ob.iou_loss_ltrb(pp, tgt, "giou") - ob.iou_loss_ltrb(pm, tgt, "giou")) / (2 * eps)
    gnum = gnum * w[:, None] * fg[:, None] / max(1.0, float(w[fg].sum()))
    got = dp.float().cpu().numpy()
    assert np.allclose(got, gnum, rtol=3e-2, atol=2e-6)
    # BCE with logits on fg rows
    x = torch.from_numpy(rng.normal(0, 2, rows).astype(np.float32)).to(torch.bfloat16)
    t = rng.uniform(0, 1, rows).astype(np.float32)
    nf = torch.tensor([float(fg.sum())], dtype=torch.float32, device="cuda")
    loss.zero_()
    dx = torch.empty((rows,), dtype=torch.bfloat16, device="cuda")
    ops.bce_logits_fwd_bwd(x.cuda(), _dev(t), _dev(lab), rows, nf, loss, dx)
    xf = x.float().numpy().astype(np.float64)
    ref = ob.binary_cross_entropy(xf[fg], t[fg]).sum() / fg.sum()
    assert abs(float(loss.item()) - ref) / ref < 1e-4
    refg = (1 / (1 + np.exp(-xf)) - t) * fg / fg.sum()
    assert np.allclose(dx.float().cpu().numpy(), refg, rtol=1e-2, atol=1e-9)


def test_atss_assign_bit_exact():
    """ATSS.get_ground_truth (models/det/atss.py:17-86): labels, ltrb offsets and centre-ness bit-exact against the oracle,
    including a gt-free image, duplicated gts (ties between gts) and gts hanging over the image border."""
    ops = _ops()
    rng = np.random.default_rng(11)
    sizes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    strides = [8, 16, 32, 64, 128]
    pts = ob.point_anchors(sizes, strides, 0.5, 1)
    allp = np.concatenate(pts, 0)
    P = allp.shape[0]
    N, Gmax = 4, 8
    gt = np.zeros((N, Gmax, 5), np.float32)
    num = np.array([6, 0, 3, 8], np.int32)
    for n in range(N):
        for g in range(num[n]):
            cx, cy = rng.uniform(0, 320), rng.uniform(0, 256)
            w, h = rng.uniform(10, 200), rng.uniform(10, 200)
            gt[n, g] = [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(1, 81)]
    gt[2, 2] = gt[2, 1]                                  # identical gts: the lower index must win
    gt[3, 0, :4] = [-30, -20, 90, 70]                    # partly outside the image
    lvl_start = [0]
    for (h, w) in sizes:
        lvl_start.append(lvl_start[-1] + h * w)
    labels = torch.empty((N, P), dtype=torch.int32, device="cuda")
    offs = torch.empty((N, P, 4), dtype=torch.float32, device="cuda")
    ctr = torch.empty((N, P), dtype=torch.float32, device="cuda")
    stats = torch.zeros((2,), dtype=torch.float32, device="cuda")
    ws = torch.empty((ops.atss_assign_workspace_bytes(N, P),), dtype=torch.uint8, device="cuda")
    ops.atss_assign(_dev(allp), lvl_start, strides, 9, 8, _dev(gt), _dev(num), labels, offs, ctr, stats, ws)
    rl, ro, rc = ob.atss_ground_truth(pts, strides, gt, num, 8, 9)
    assert np.array_equal(labels.cpu().numpy(), rl)
    assert np.array_equal(offs.cpu().numpy(), ro)
    got_c = ctr.cpu().numpy()
    fg = rl > 0
    assert fg.sum() > 20 and np.array_equal(got_c[fg], rc[fg])
    assert np.array_equal(np.nan_to_num(got_c), np.nan_to_num(rc))
    st = stats.cpu().numpy()
    assert st[0] == fg.sum() and abs(st[1] - rc[fg].sum()) <= 1e-4 * rc[fg].sum()
