"""CPU suite: the oracle against the reference's own golden vectors, and the C-ABI surface (no compute calls)."""
import os
import re

import numpy as np

from oracle import box_ops as ob


def test_oracle_matches_reference_box_kats(golden_dir):
    """tests/structures/test_boxes.py:38-86 of the reference."""
    k = np.load(os.path.join(golden_dir, "reference_kat.npz"))
    assert np.allclose(ob.box_iou(k["boxes1"], k["boxes2"]), k["iou_1x2"])
    assert np.allclose(ob.box_ioa(k["boxes2"], k["boxes1"]), k["ioa_2x1"])
    assert np.allclose(ob.intersection(k["boxes1"], k["boxes2"]), k["inter_1x2"])
    assert np.allclose(ob.box_centers(k["boxes1"]), k["centers_1"])
    assert np.allclose(ob.box_scale(k["boxes1"], 2), k["boxes1"] * 2)


def test_oracle_matches_reference_nms_kat(golden_dir):
    """tests/layers/test_postprocess.py:13-28: keep == [0, 3, 4, 2]."""
    k = np.load(os.path.join(golden_dir, "reference_kat.npz"))
    keep = ob.batched_nms(k["nms_boxes"], k["nms_scores"], k["nms_labels"], float(k["nms_iou_thresh"]))
    assert keep.tolist() == k["nms_keep"].tolist()


def test_oracle_nms_matches_reference_py_cpu_nms(golden_dir):
    """tests/golden/reference_nms.npz: keep lists of the reference's own numpy NMS (layers/common/post_processing.py:106-132) on seeded
    random boxes with distinct scores, duplicates and zero-area boxes -- pins the suppression rule (IoU > thr) and the IoU form."""
    k = np.load(os.path.join(golden_dir, "reference_nms.npz"))
    for i in k["cases"]:
        keep = ob.nms(k[f"boxes_{i}"], k[f"scores_{i}"], float(k[f"thr_{i}"]))
        assert keep.tolist() == k[f"keep_{i}"].tolist(), int(i)
        # the class-offset form with a single class is the same problem
        keep = ob.batched_nms(k[f"boxes_{i}"], k[f"scores_{i}"], np.zeros(len(k[f"scores_{i}"]), np.int32), float(k[f"thr_{i}"]))
        assert keep.tolist() == k[f"keep_{i}"].tolist(), int(i)


def test_oracle_matches_reference_pad_kat(golden_dir):
    """tests/layers/test_preprocess.py:13-35: padded shapes and sum preservation."""
    k = np.load(os.path.join(golden_dir, "reference_kat.npz"))
    for shp, rank, hw in zip(k["pad_in_shapes"], k["pad_in_rank"], k["pad_out_hw"]):
        shape = tuple(int(v) for v in shp[-int(rank):])
        data = np.ones(shape, np.float32)
        out = ob.get_padded_tensor(data)
        assert out.shape == shape[:-2] + (int(hw[0]), int(hw[1]))
        assert out.sum() == data.sum()


def test_dummy_loader_fixture(golden_dir):
    """Captured from basedet/utils/dummy.py (tests/golden/make_golden.py)."""
    d = np.load(os.path.join(golden_dir, "dummy_loader.npz"))
    assert d["anno_800x1344"].shape == (2, 10, 5) and d["im_info_800x1344"].tolist() == [[800, 1344, 612, 612, 10], [800, 1344, 500, 375, 5]]
    assert np.allclose(d["anno_512x512"], d["anno_800x1344"] * np.float32(0.64))
    assert np.array_equal(d["batch2_gt_boxes"], ob.tile_batch(d["anno_800x1344"], 2))
    b16 = ob.tile_batch(d["anno_800x1344"], 16)
    assert b16.shape == (16, 10, 5) and np.array_equal(b16[7], d["anno_800x1344"][0]) and np.array_equal(b16[8], d["anno_800x1344"][1])


def test_matcher_and_coder_properties():
    rng = np.random.default_rng(0)
    m = rng.uniform(0, 1, (6, 500)).astype(np.float32)
    idx, lab = ob.matcher(m, [0.4, 0.5], [0, -1, 1], True)
    mx = m.max(0)
    assert np.array_equal(idx, m.argmax(0))
    assert set(np.unique(lab)) <= {-1, 0, 1}
    assert np.all(lab[mx >= 0.5] == 1) and np.all(lab[m.argmax(1)] == 1)   # low-quality: every gt keeps its best anchor
    xy = rng.uniform(0, 100, (50, 2)).astype(np.float32); wh = rng.uniform(5, 50, (50, 2)).astype(np.float32)
    a = np.concatenate([xy, xy + wh], 1)
    xy = rng.uniform(0, 100, (50, 2)).astype(np.float32); wh = rng.uniform(5, 50, (50, 2)).astype(np.float32)
    g = np.concatenate([xy, xy + wh], 1)
    assert np.allclose(ob.box_decode(a, ob.box_encode(a, g)), g, atol=1e-3)
    assert np.allclose(ob.box_encode(a, a), 0)


def test_focal_grad_matches_numeric():
    rng = np.random.default_rng(1)
    x = rng.normal(0, 2, (50, 8)); t = (rng.uniform(size=(50, 8)) < 0.2).astype(np.float64)
    e = 1e-6
    num = (ob.sigmoid_focal_loss(x + e, t, 0.25, 2.0) - ob.sigmoid_focal_loss(x - e, t, 0.25, 2.0)) / (2 * e)
    assert np.allclose(ob.sigmoid_focal_loss_grad(x, t, 0.25, 2.0), num, rtol=1e-5, atol=1e-8)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads on a CPU-only box and exports everything include/basedet_hip.h declares."""
    from basedet_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "basedet_hip.h")).read()
    declared = set(re.findall(r"\b(bd_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.bd_version() >= 100


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from basedet_amd import _lib
    with pytest.raises(_lib.BasedetHipError):
        _lib.ptr(torch.zeros(4))
