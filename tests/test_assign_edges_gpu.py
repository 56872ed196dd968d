"""Target assignment on the reference's edge cases, bit-exact against the oracle at 800 x 1344 (201 600 anchors / 22 400 points):

* G in {1, 37, 100} valid gts per image (the DummyLoader pattern never exceeds 10);
* zero-area gts, gts completely outside the image, exact duplicates (argmax / argmin ties -> lowest index), a gt whose row-max IoU is
  0 (layers/common/matcher.py:47-49: `matrix == max(matrix, axis=1)` then marks EVERY zero-IoU anchor positive -- SURVEY a10's edge);
* num_gt < Gmax with NaN / huge garbage in the padding rows (models/det/retinanet.py:216 slices `[:num_boxes]`: they must never be read).

Kernels: bd_retina_assign_encode, bd_fcos_assign, bd_atss_assign, bd_rpn_assign_encode (+ bd_sample_labels), bd_rcnn_sample_targets."""
import numpy as np
import pytest
import torch

from oracle import box_ops as ob
from oracle import rcnn_ops as orc

pytestmark = pytest.mark.gpu

H, W = 800, 1344
STRIDES = [8, 16, 32, 64, 128]
SIZES = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
SCALES = [[32, 40.31747359663594, 50.79683366298238], [64, 80.63494719327188, 101.59366732596476],
          [128, 161.26989438654377, 203.18733465192952], [256, 322.53978877308754, 406.37466930385904],
          [512, 645.0795775461751, 812.7493386077181]]


def _ops():
    from basedet_amd import ops
    return ops


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def edge_gts(rng, counts, Gmax, kinds=("zero_area", "outside", "dup", "rowmax0"), garbage=True):
    """(N, Gmax, 5) gt rows + num_gt.  Image i holds counts[i] valid boxes; the special kinds are planted into the images that have
    room for them (one kind per slot from the end), the padding rows carry NaN / 1e30 garbage."""
    N = len(counts)
    gt = np.zeros((N, Gmax, 5), np.float32)
    for n, G in enumerate(counts):
        cx, cy = rng.uniform(0, W, G), rng.uniform(0, H, G)
        w, h = rng.uniform(8, 500, G), rng.uniform(8, 400, G)
        b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        b = np.round(b * 4) / 4                             # quarter-pixel coordinates: exact ties between anchors and gts do occur
        gt[n, :G, :4] = b
        gt[n, :G, 4] = rng.integers(1, 81, G)
        if G >= 8:
            s = G - 1
            for k in kinds:
                # a zero-area or unreachable gt has row-max IoU 0 and (low-quality rule) turns EVERY anchor positive, which would hide
                # the other cases: image 0 of a batch gets neither, the later images one of them each
                if k == "zero_area" and n % 3 == 1:
                    gt[n, s, :4] = [300.0, 200.0, 300.0, 260.0]                  # x1 == x2
                elif k == "outside":
                    gt[n, s, :4] = [W + 50.0, H + 20.0, W + 300.0, H + 200.0]    # overlaps only anchors hanging over the border
                elif k == "dup":
                    gt[n, s] = gt[n, 0]
                    gt[n, s, 4] = (int(gt[n, 0, 4]) % 80) + 1                    # same box, another class: the LOWER index must win
                elif k == "rowmax0" and n % 3 == 2:
                    gt[n, s, :4] = [5000.0, 5000.0, 5040.0, 5030.0]              # no anchor reaches it: row-max IoU = 0
                else:
                    continue
                s -= 1
        if garbage and G < Gmax:
            gt[n, G:] = np.where(rng.random((Gmax - G, 5)) < 0.5, np.nan, 1e30).astype(np.float32)
    return gt, np.asarray(counts, np.int32)


def _anchors():
    return np.concatenate(ob.default_anchors(SIZES, STRIDES, SCALES, [[0.5, 1, 2]] * 5, 0.5), 0).astype(np.float32)


@pytest.mark.parametrize("counts,Gmax", [([1, 1], 1), ([37, 5, 0, 37, 12], 40), ([100, 63, 100], 100)])
def test_retina_assign_edge_cases(counts, Gmax):
    ops = _ops()
    rng = np.random.default_rng(100 + Gmax)
    anchors = _anchors()
    A = anchors.shape[0]
    assert A == 201600
    gt, ng = edge_gts(rng, counts, Gmax)
    N = len(counts)
    labels = torch.empty((N, A), dtype=torch.int32, device="cuda")
    midx = torch.empty((N, A), dtype=torch.int32, device="cuda")
    offs = torch.empty((N, A, 4), dtype=torch.float32, device="cuda")
    nfg = torch.zeros((1,), dtype=torch.int32, device="cuda")
    ws = torch.empty((N * Gmax,), dtype=torch.float32, device="cuda")
    for thr, lq in (((0.4, 0.5), True), ((0.4, 0.5), False), ((0.3, 0.7), True)):
        nfg.zero_()
        ops.retina_assign_encode(_dev(anchors), _dev(gt), _dev(ng), thr[0], thr[1], lq, (0, 0, 0, 0), (1, 1, 1, 1), labels, midx, offs, nfg, ws)
        rl, ro, ri = ob.retinanet_ground_truth(anchors, gt, ng, thresholds=thr, allow_low_quality=lq)
        gl = labels.cpu().numpy()
        assert np.array_equal(gl, rl), (thr, lq, int((gl != rl).sum()))
        assert np.array_equal(midx.cpu().numpy(), ri)
        assert int(nfg.item()) == int((rl > 0).sum())
        go = offs.cpu().numpy()
        fin = np.isfinite(ro).all(axis=-1) & (rl > 0)          # log(0 / aw) of a zero-width gt is -inf on both sides: compare the finite rows
        assert np.array_equal(np.isfinite(go).all(axis=-1) & (rl > 0), fin)
        assert np.array_equal(go[fin][:, :2], ro[fin][:, :2])
        assert np.allclose(go[fin][:, 2:], ro[fin][:, 2:], rtol=2e-6, atol=2e-6)
    if Gmax == 100:
        # the zero-area gt of image 1 and the unreachable gt of image 2 have row-max IoU 0: every anchor with IoU 0 (all of them) is positive
        rl, _, _ = ob.retinanet_ground_truth(anchors, gt, ng)
        assert (rl[1] > 0).all() and (rl[2] > 0).all() and 0 < (rl[0] > 0).sum() < 20000 and (rl[0] == -1).sum() > 0


@pytest.mark.parametrize("counts,Gmax", [([1, 1], 1), ([37, 5, 0, 37, 12], 40), ([100, 63, 100], 100)])
def test_fcos_and_atss_assign_edge_cases(counts, Gmax):
    ops = _ops()
    rng = np.random.default_rng(200 + Gmax)
    gt, ng = edge_gts(rng, counts, Gmax)
    N = len(counts)
    pts = ob.point_anchors(SIZES, STRIDES, 0.5, 1)
    allp = np.concatenate(pts).astype(np.float32)
    P = allp.shape[0]
    assert P == 22400
    start = np.cumsum([0] + [p.shape[0] for p in pts]).tolist()
    soi = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, float("inf")]]
    labels = torch.empty((N, P), dtype=torch.int32, device="cuda")
    offs = torch.empty((N, P, 4), dtype=torch.float32, device="cuda")
    ctr = torch.empty((N, P), dtype=torch.float32, device="cuda")
    stats = torch.zeros((2,), dtype=torch.float32, device="cuda")
    for radius in (1.5, 0.0):
        stats.zero_()
        ops.fcos_assign(_dev(allp), start, soi, STRIDES, radius, _dev(gt), _dev(ng), labels, offs, ctr, stats)
        rl, ro, rc = ob.fcos_ground_truth(pts, STRIDES, gt, ng, soi, radius)
        assert np.array_equal(labels.cpu().numpy(), rl), radius
        assert np.array_equal(offs.cpu().numpy(), ro)
        fg = rl > 0
        assert np.array_equal(ctr.cpu().numpy()[fg], rc[fg])
        st = stats.cpu().numpy()
        assert st[0] == fg.sum() and np.isclose(st[1], rc[fg].sum(), rtol=1e-5)
    # ATSS on the same gts (models/det/atss.py:17-86)
    ws = torch.empty((ops.atss_assign_workspace_bytes(N, P),), dtype=torch.uint8, device="cuda")
    stats.zero_()
    ops.atss_assign(_dev(allp), start, STRIDES, 9, 8, _dev(gt), _dev(ng), labels, offs, ctr, stats, ws)
    rl, ro, rc = ob.atss_ground_truth(pts, STRIDES, gt, ng, 8, 9)
    assert np.array_equal(labels.cpu().numpy(), rl)
    assert np.array_equal(offs.cpu().numpy(), ro)
    fg = rl > 0
    assert np.array_equal(ctr.cpu().numpy()[fg], rc[fg])
    assert stats.cpu().numpy()[0] == fg.sum()


@pytest.mark.parametrize("counts,Gmax", [([1, 1], 1), ([37, 0, 9], 40), ([100, 100, 64], 100)])
def test_rpn_assign_and_sampling_edge_cases(counts, Gmax):
    """RPN.get_ground_truth (models/det/rpn.py:215-240) over the P2-P6 anchors of 800 x 1344 (268 569), Matcher(0.3, 0.7) + subsampling."""
    ops = _ops()
    rng = np.random.default_rng(300 + Gmax)
    sizes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    strides = [4, 8, 16, 32, 64]
    anchors = np.concatenate(ob.default_anchors(sizes, strides, [[32], [64], [128], [256], [512]], [[0.5, 1, 2]] * 5, 0.5), 0).astype(np.float32)
    A = anchors.shape[0]
    assert A == 268569
    gt, ng = edge_gts(rng, counts, Gmax)
    N = len(counts)
    kp = rng.random((N, A), dtype=np.float32)
    kn = (np.round(rng.random((N, A), dtype=np.float32) * 4096) / 4096).astype(np.float32)
    labels = torch.empty((N, A), dtype=torch.int32, device="cuda")
    match = torch.empty((N, A), dtype=torch.int32, device="cuda")
    offs = torch.empty((N, A, 4), dtype=torch.float32, device="cuda")
    nfg = torch.zeros((1,), dtype=torch.int32, device="cuda")
    nvalid = torch.zeros((1,), dtype=torch.int32, device="cuda")
    ws = torch.empty((N * Gmax,), dtype=torch.float32, device="cuda")
    ops.rpn_assign_encode(_dev(anchors), _dev(gt), _dev(ng), 0.3, 0.7, True, [0, 0, 0, 0], [1, 1, 1, 1], labels, match, offs, nfg, ws)
    ops.sample_labels(labels, _dev(kp), _dev(kn), 128, 256, nvalid)
    ref_l, ref_o = orc.rpn_ground_truth(anchors, gt, ng, kp, kn, (0.3, 0.7), (0, -1, 1), True, 256, 128)
    gl = labels.cpu().numpy()
    assert np.array_equal(gl, ref_l), int((gl != ref_l).sum())
    assert int(nvalid.item()) == int((ref_l >= 0).sum())
    fg = (ref_l > 0) & np.isfinite(ref_o).all(axis=-1)
    np.testing.assert_allclose(offs.cpu().numpy()[fg], ref_o[fg], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("counts,Gmax", [([1, 1], 1), ([37, 0, 9], 40), ([100, 100, 64], 100)])
def test_rcnn_sample_targets_edge_cases(counts, Gmax):
    """RCNN.get_ground_truth (layers/head/rcnn.py:95-147): 1000 proposals + the gts appended, IoU / argmax / fg-bg bands / key-ordered
    sampling of 512 with duplicates, zero-area and out-of-image gts; proposals include zero-area boxes and exact copies of gts."""
    ops = _ops()
    rng = np.random.default_rng(400 + Gmax)
    gt, ng = edge_gts(rng, counts, Gmax, kinds=("zero_area", "outside", "dup"))
    N, post_k = len(counts), 1000
    rois = np.zeros((N, post_k, 4), np.float32)
    num_rois = np.full((N,), post_k, np.int32)
    num_rois[-1] = 417
    for n in range(N):
        m = int(num_rois[n])
        cx, cy = rng.uniform(0, W, m), rng.uniform(0, H, m)
        w, h = rng.uniform(4, 400, m), rng.uniform(4, 400, m)
        b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        g = int(ng[n])
        if g:
            k = m // 2
            b[:k] = gt[n, rng.integers(0, g, k), :4] + rng.normal(0, 5, (k, 4)).astype(np.float32)
            b[k: k + min(g, 20)] = gt[n, : min(g, 20), :4]                      # exact copies: IoU 1 with a gt (and with its duplicate)
        b[-3:, 2] = b[-3:, 0]                                                   # zero-area proposals
        b[:, 0::2] = np.clip(b[:, 0::2], 0, W); b[:, 1::2] = np.clip(b[:, 1::2], 0, H)
        rois[n, :m] = b
    key_ld = post_k + Gmax
    kf = rng.random((N, key_ld), dtype=np.float32)
    kb = (np.round(rng.random((N, key_ld), dtype=np.float32) * 256) / 256).astype(np.float32)
    std = [0.1, 0.1, 0.2, 0.2]
    S, nfgmax = 512, 256
    o_rois = torch.empty((N, S, 4), dtype=torch.float32, device="cuda")
    o_lab = torch.empty((N, S), dtype=torch.int32, device="cuda")
    o_tgt = torch.empty((N, S, 4), dtype=torch.float32, device="cuda")
    o_cnt = torch.empty((N,), dtype=torch.int32, device="cuda")
    tot = torch.zeros((1,), dtype=torch.int32, device="cuda")
    gt_clean = np.where(np.isfinite(gt), gt, 0).astype(np.float32)               # the oracle gets the sliced rows; the kernel the garbage too
    ops.rcnn_sample_targets(_dev(rois), _dev(num_rois), _dev(gt), _dev(ng), _dev(kf), _dev(kb), S, nfgmax, 0.5, 0.5, 0.0,
                            [0, 0, 0, 0], std, o_rois, o_lab, o_tgt, o_cnt, tot)
    for n in range(N):
        rr, rl, rt = orc.rcnn_ground_truth(rois[n, : num_rois[n]], gt_clean[n, : ng[n]], kf[n], kb[n], S, 0.5, 0.5, 0.5, 0.0, (0, 0, 0, 0), std)
        m = len(rl)
        assert int(o_cnt[n].item()) == m
        gl = o_lab[n].cpu().numpy()
        assert np.array_equal(gl[:m], rl), (n, int((gl[:m] != rl).sum()))
        assert np.all(gl[m:] == -1)
        assert np.array_equal(o_rois[n].cpu().numpy()[:m], rr)
        fin = np.isfinite(rt).all(axis=-1) & (rl > 0)
        np.testing.assert_allclose(o_tgt[n].cpu().numpy()[:m][fin], rt[fin], rtol=2e-5, atol=2e-5)
