"""Batch contract on the host side of the hot path: DetectionPadCollator (data/collators/pad_collator.py:23-61) and the
aspect-ratio grouped sampler (data/samplers/group_sampler.py:8-93)."""
import numpy as np

from basedet_amd.data import AspectRatioGroupSampler, DetectionPadCollator, calculate_padding_shape


def test_calculate_padding_shape():
    assert calculate_padding_shape((3, 5, 7), (3, 8, 7)) == ((0, 0), (0, 3), (0, 0))


def test_pad_collator_contract():
    rng = np.random.default_rng(0)
    img0 = rng.integers(0, 255, (3, 20, 31)).astype(np.uint8)
    img1 = rng.integers(0, 255, (3, 24, 17)).astype(np.uint8)
    boxes0 = np.array([[1, 2, 10, 12], [3, 3, 8, 9], [0, 0, 5, 5]], np.float64)
    boxes1 = np.array([[2, 2, 9, 9]], np.float64)
    out = DetectionPadCollator().apply([(img0, boxes0, np.array([5, 7, 1]), (40, 62, 0)), (img1, boxes1, np.array([80]), (48, 34, 0))])
    assert set(out) == {"data", "gt_boxes", "im_info"}
    assert out["data"].shape == (2, 3, 24, 31) and out["data"].dtype == np.float32
    assert np.array_equal(out["data"][0, :, :20, :], img0.astype(np.float32)) and not out["data"][0, :, 20:, :].any()
    assert np.array_equal(out["data"][1, :, :, :17], img1.astype(np.float32)) and not out["data"][1, :, :, 17:].any()
    assert out["gt_boxes"].shape == (2, 3, 5)
    assert np.array_equal(out["gt_boxes"][0], np.concatenate([boxes0, [[5], [7], [1]]], 1).astype(np.float32))
    assert np.array_equal(out["gt_boxes"][1, 0], [2, 2, 9, 9, 80]) and not out["gt_boxes"][1, 1:].any()
    assert np.array_equal(out["im_info"], np.array([[20, 31, 40, 62, 3], [24, 17, 48, 34, 1]], np.float32))
    # an image without boxes keeps the (0, 5) row layout
    out = DetectionPadCollator(pad_value=-1.0).apply([(img0, np.zeros((0, 4)), np.zeros((0,)), (20, 31)), (img1, boxes1, np.array([3]), (24, 17))])
    assert out["gt_boxes"].shape == (2, 1, 5) and np.all(out["gt_boxes"][0] == -1) and out["im_info"][0, 4] == 0
    assert np.all(out["data"][1, :, :, 17:] == -1)


class _DS:
    def __init__(self, hw):
        self.hw = hw

    def __len__(self):
        return len(self.hw)

    def get_img_info(self, i):
        return {"height": self.hw[i][0], "width": self.hw[i][1]}


def test_aspect_ratio_group_sampler():
    hw = [(480, 640)] * 13 + [(640, 480)] * 11 + [(500, 500)] * 3      # landscape / portrait / square (ratio 1 -> portrait group)
    ds = _DS(hw)
    s = AspectRatioGroupSampler(ds, 4, seed=3)
    batches = [b for b in s]          # (list(s) would consult __len__, which the reference leaves undefined)
    assert len(batches) == (13 // 4) + (14 // 4)
    for b in batches:
        assert len(b) == 4 and len({hw[i][0] < hw[i][1] for i in b}) == 1
    assert len({i for b in batches for i in b}) == sum(len(b) for b in batches)
    # two ranks see disjoint index sets that together cover one padded permutation
    r0 = AspectRatioGroupSampler(ds, 2, seed=5, world_size=2, rank=0)
    r1 = AspectRatioGroupSampler(ds, 2, seed=5, world_size=2, rank=1)
    i0 = {i for b in r0 for i in b}; i1 = {i for b in r1 for i in b}
    assert not (i0 & i1) or len(i0 & i1) <= 1          # at most the wrap-around padding element is shared


# ---- fixtures produced by the reference's OWN code (tests/golden/make_golden.py batch_contract_fixtures: the bodies of
# DetectionPadCollator.apply / GroupedRandomSampler.batch and the helper closures of AspectRatioGroupSampler.__init__ run in the build
# container from the files where they lie; arrays only are committed) --------------------------------------------------------------------
import os

_GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_batch_contract.npz"))


def test_calculate_padding_shape_matches_reference_outputs():
    for i in range(int(_GOLD["pad_n"])):
        got = calculate_padding_shape(tuple(_GOLD[f"pad_orig_{i}"].tolist()), tuple(_GOLD[f"pad_target_{i}"].tolist()))
        assert np.array_equal(np.asarray(got, np.int64).reshape(-1, 2), _GOLD[f"pad_out_{i}"].reshape(-1, 2))


def test_pad_collator_matches_reference_outputs():
    for c in range(int(_GOLD["collate_n"])):
        inputs = [(_GOLD[f"collate_{c}_img_{k}"], _GOLD[f"collate_{c}_boxes_{k}"], _GOLD[f"collate_{c}_cat_{k}"],
                   tuple(_GOLD[f"collate_{c}_info_{k}"].tolist())) for k in range(int(_GOLD[f"collate_{c}_count"]))]
        out = DetectionPadCollator(pad_value=float(_GOLD[f"collate_{c}_pad_value"])).apply(inputs)
        assert set(out) == {"data", "gt_boxes", "im_info"}
        for key in ("data", "gt_boxes", "im_info"):
            want = _GOLD[f"collate_{c}_out_{key}"]
            assert out[key].dtype == want.dtype == np.float32 and out[key].shape == want.shape, (c, key)
            assert np.array_equal(out[key], want), (c, key)          # bit for bit: the contract is copies and exact casts


class _HW:
    def __init__(self, hw):
        self.hw = hw

    def __len__(self):
        return len(self.hw)

    def get_img_info(self, i):
        return {"height": int(self.hw[i][0]), "width": int(self.hw[i][1])}


def test_aspect_ratio_groups_match_reference_outputs():
    ds = _HW(_GOLD["sampler_hw"])
    for b in range(int(_GOLD["sampler_bins_n"])):
        s = AspectRatioGroupSampler(ds, 2, aspect_grouping=_GOLD[f"sampler_bins_{b}"].tolist())
        assert np.array_equal(np.asarray(s.group_ids, np.int64), _GOLD[f"sampler_groups_{b}"]), b
    # the ratios themselves (height / width in float64)
    got = np.array([h / w for h, w in _GOLD["sampler_hw"].tolist()], np.float64)
    assert np.array_equal(got, _GOLD["sampler_ratios"])


def test_grouped_batches_match_reference_outputs_over_three_passes():
    """Same permutations in, same batches out -- including what the unfilled group buffers carry from one pass into the next."""
    ds = _HW(_GOLD["sampler_hw"])
    for r in range(int(_GOLD["sampler_runs_n"])):
        bs = int(_GOLD[f"sampler_run_{r}_batch_size"])
        s = AspectRatioGroupSampler(ds, bs, aspect_grouping=_GOLD[f"sampler_run_{r}_bins"].tolist())
        for e in range(3):
            perm = _GOLD[f"sampler_run_{r}_perm_{e}"]
            s.sample = lambda perm=perm: perm
            got = [list(b) for b in s.batch()]
            assert len(got) == int(_GOLD[f"sampler_run_{r}_count_{e}"]) and all(len(b) == bs for b in got), (r, e)
            assert [i for b in got for i in b] == _GOLD[f"sampler_run_{r}_flat_{e}"].tolist(), (r, e)
