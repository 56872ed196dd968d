"""Name-matching rules of the reference's checkpoint loader (basedet/utils/checkpoint.py:41-140), CPU only."""
import numpy as np
import pytest

from basedet_amd.utils.checkpoint import full_match, load_checkpoint_file, load_matched_weights, save_checkpoint, unwarp_ckpt


class _FakeModel:
    def __init__(self):
        self.p = {"backbone.bottom_up.conv1.weight": np.zeros((4, 3, 7, 7), np.float32),
                  "backbone.bottom_up.bn1.weight": np.ones((4,), np.float32),
                  "head.conv.weight": np.zeros((2, 4, 3, 3), np.float32),
                  "head.cls.weight": np.zeros((5, 4, 3, 3), np.float32)}

    def state_dict(self):
        return {k: v.copy() for k, v in self.p.items()}

    def _bind_params(self, state):
        self.p = state


def test_full_match_rules():
    shapes = {"backbone.bottom_up.conv1.weight": (4, 3, 7, 7), "head.conv1.weight": (4, 3, 7, 7), "head.fc.weight": (10, 4)}
    w = {"backbone.bottom_up.conv1.weight": np.zeros((4, 3, 7, 7)), "fc.weight": np.zeros((10, 4)), "other": np.zeros(3)}
    mapping, unused = full_match(w, shapes)
    assert mapping == {"backbone.bottom_up.conv1.weight": "backbone.bottom_up.conv1.weight", "head.fc.weight": "fc.weight"}
    assert unused == ["other"]
    # ambiguous suffix resolved by the element count
    shapes = {"a.conv.weight": (2, 2), "b.conv.weight": (3, 3)}
    mapping, _ = full_match({"conv.weight": np.zeros((9,))}, shapes)
    assert mapping == {"b.conv.weight": "conv.weight"}


def test_load_backbone_checkpoint_roundtrip(tmp_path):
    m = _FakeModel()
    rng = np.random.default_rng(0)
    # an ImageNet-style backbone file: keys without the "backbone.bottom_up." prefix, BN vector dumped as (1, C, 1, 1)
    ck = {"conv1.weight": rng.normal(size=(4, 3, 7, 7)).astype(np.float32), "bn1.weight": rng.normal(size=(1, 4, 1, 1)).astype(np.float32),
          "fc.weight": np.zeros((10, 4), np.float32)}
    path = tmp_path / "r.pkl"
    save_checkpoint(path, ck, epoch=3)
    raw = load_checkpoint_file(path)
    assert raw["epoch"] == 3 and set(unwarp_ckpt(raw)) == set(ck)
    load_matched_weights(m, str(path))
    assert np.array_equal(m.p["backbone.bottom_up.conv1.weight"], ck["conv1.weight"])
    assert np.array_equal(m.p["backbone.bottom_up.bn1.weight"], ck["bn1.weight"].reshape(4))
    assert not m.p["head.conv.weight"].any()
    np.savez(tmp_path / "r.npz", **{"head.conv.weight": np.ones((2, 4, 3, 3), np.float32), "head.cls.weight": np.ones((7, 4, 3, 3), np.float32)})
    load_matched_weights(m, str(tmp_path / "r.npz"))
    assert m.p["head.conv.weight"].all() and not m.p["head.cls.weight"].any()      # shape mismatch skipped when not strict
    with pytest.raises(ValueError):
        load_matched_weights(m, str(tmp_path / "r.npz"), strict=True)


def test_pickle_with_foreign_objects_is_refused(tmp_path):
    import pickle
    path = tmp_path / "bad.pkl"
    with open(path, "wb") as f:
        pickle.dump({"x": pytest}, f) if False else f.write(pickle.dumps({"x": slice(1, 2)}))
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_file(path)


def test_full_match_reproduces_the_reference_outputs(golden_dir):
    """tests/golden/reference_checkpoint_match.json: outputs of the reference's own get_name_matched_keys / get_shape_matched_keys /
    full_match (utils/checkpoint.py:13-90), generated in the build container by tests/golden/make_golden.py."""
    import json
    import os
    from basedet_amd.utils.checkpoint import _name_matched
    with open(os.path.join(golden_dir, "reference_checkpoint_match.json")) as f:
        d = json.load(f)
    model = {k: tuple(v) for k, v in d["model"]}
    for q, want in d["name_matched"].items():
        assert sorted(_name_matched(q, model.keys())) == want, q
    for name, rec in d["cases"].items():
        w = {k: np.zeros(tuple(v), np.float32) for k, v in rec["weights"]}
        if rec.get("raises"):
            with pytest.raises(AssertionError):
                full_match(w, model)
            continue
        mapping, unused = full_match(w, model)
        assert mapping == rec["mapping"], name
        assert unused == rec["unused"], name
    # the ImageNet-backbone case end to end: BatchNorm vectors dumped as (1, C, 1, 1) are reshaped on load (checkpoint.py:24-28, 118-121)
    rec = d["cases"]["imagenet_backbone"]

    class M:
        def __init__(self):
            self.p = {k: np.zeros(v, np.float32) for k, v in model.items()}

        def state_dict(self):
            return {k: v.copy() for k, v in self.p.items()}

        def _bind_params(self, st):
            self.p = st

    rng = np.random.default_rng(0)
    ck = {k: rng.normal(size=tuple(v)).astype(np.float32) for k, v in rec["weights"]}
    m = M()
    load_matched_weights(m, {"model": {"state_dict": ck}})
    for mk, ckk in rec["mapping"].items():
        assert np.array_equal(m.p[mk], ck[ckk].reshape(model[mk])), mk
    assert not m.p["head.cls_score.weight"].any()
