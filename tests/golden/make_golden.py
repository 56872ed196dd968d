#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

Two kinds of vectors are captured:
  1. arrays produced by *importing reference Python by file path* (the only importable piece is
     basedet/utils/dummy.py, numpy-only): DummyLoader.anno / im_info and one tiled batch.
  2. known-answer vectors held by the reference's own unit tests, re-typed here as DATA
     (inputs + expected outputs), each with the test file:line it comes from.
The reference sources themselves never enter this repository.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("BASEDET_REFERENCE", "/root/reference")


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def functions_by_name(path, names, env):
    """The named top-level (pure numpy / Python) functions of a reference file that cannot be imported as a module (its import block
    needs megengine / loguru): their definitions are compiled from the file where it lies and run in `env`.  Build container only;
    nothing of the source is written anywhere -- only the arrays the functions return."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in picked} == set(names), (path, names)
    ns = dict(env)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def methods_by_name(path, cls, names, env, nested_in=None):
    """The named METHODS of class `cls` in a reference file whose module (or whose base class) cannot be imported: their definitions are
    lifted out of the class body and compiled as plain functions taking `self` (the caller passes a stub object carrying the attributes
    the body reads).  With `nested_in`, the named functions are instead the ones defined INSIDE that method's body (the helper closures
    of AspectRatioGroupSampler.__init__).  Build container only; only the arrays the functions return are written anywhere."""
    import ast
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    (cdef,) = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls]
    body = cdef.body
    if nested_in is not None:
        (outer,) = [n for n in body if isinstance(n, ast.FunctionDef) and n.name == nested_in]
        body = outer.body
    picked = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in picked} == set(names), (path, cls, names)
    ns = dict(env)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def batch_contract_fixtures():
    """Outputs of the reference's own batch-contract code on seeded inputs:
      * calculate_padding_shape and DetectionPadCollator.apply (data/collators/pad_collator.py:15-61): the method body runs with a stub
        `self` that carries pad_value (its base class, megengine.data.Collator, is not installed);
      * GroupedRandomSampler.batch (data/samplers/group_sampler.py:39-54): the grouping arithmetic -- an index joins its group's buffer,
        a batch leaves when the buffer holds batch_size entries, unfilled buffers survive into the next pass -- on GIVEN permutations
        (sample() / scatter() belong to megengine's RandomSampler and are inputs here), two passes in a row;
      * the helper closures of AspectRatioGroupSampler.__init__ (:83-92): height / width ratios and their bisect-right group ids."""
    import bisect
    import types
    from collections import defaultdict
    cpath = os.path.join(REF, "basedet/data/collators/pad_collator.py")
    (cps,) = functions_by_name(cpath, ["calculate_padding_shape"], {})
    (apply,) = methods_by_name(cpath, "DetectionPadCollator", ["apply"], {"np": np, "defaultdict": defaultdict, "calculate_padding_shape": cps})
    rng = np.random.default_rng(77)
    out = {}
    pads = [((3, 5, 7), (3, 8, 7)), ((2, 5), (4, 5)), ((3, 24, 31), (3, 24, 31)), ((0, 5), (3, 5)), ((7,), (9,))]
    out["pad_n"] = np.int32(len(pads))
    for i, (o, t) in enumerate(pads):
        out[f"pad_orig_{i}"], out[f"pad_target_{i}"] = np.asarray(o, np.int64), np.asarray(t, np.int64)
        out[f"pad_out_{i}"] = np.asarray(cps(o, t), np.int64)
    cases = [  # (pad_value, [(H, W, boxes, image dtype)])
        (0.0, [(20, 31, 3, np.uint8), (24, 17, 1, np.uint8)]),
        (-1.0, [(20, 31, 0, np.uint8), (24, 17, 2, np.float64)]),                 # an image without boxes
        (0.0, [(32, 32, 4, np.float32)]),                                         # a batch of one
        (114.0, [(17, 40, 2, np.uint8), (40, 17, 5, np.uint8), (33, 33, 1, np.float32), (8, 8, 3, np.uint8)]),
    ]
    out["collate_n"] = np.int32(len(cases))
    for c, (pv, imgs) in enumerate(cases):
        inputs = []
        for k, (H, W, g, dt) in enumerate(imgs):
            img = rng.integers(0, 255, (3, H, W)).astype(dt) if dt == np.uint8 else rng.uniform(0, 255, (3, H, W)).astype(dt)
            xy = rng.uniform(0, 10, (g, 2)); wh = rng.uniform(1, 12, (g, 2))
            boxes = np.concatenate([xy, xy + wh], 1)                               # float64, as a reader hands them over
            cat = rng.integers(1, 81, (g,)).astype(np.int64)
            info = (int(H * 1.7), int(W * 1.7), k)
            inputs.append((img, boxes, cat, info))
            out[f"collate_{c}_img_{k}"], out[f"collate_{c}_boxes_{k}"], out[f"collate_{c}_cat_{k}"] = img, boxes, cat
            out[f"collate_{c}_info_{k}"] = np.asarray(info, np.int64)
        res = apply(types.SimpleNamespace(pad_value=pv), inputs)
        assert sorted(res.keys()) == ["data", "gt_boxes", "im_info"]
        out[f"collate_{c}_pad_value"], out[f"collate_{c}_count"] = np.float64(pv), np.int32(len(imgs))
        for key in ("data", "gt_boxes", "im_info"):
            out[f"collate_{c}_out_{key}"] = res[key]
            assert res[key].dtype == np.float32
    spath = os.path.join(REF, "basedet/data/samplers/group_sampler.py")
    (batch,) = methods_by_name(spath, "GroupedRandomSampler", ["batch"], {"np": np})
    car, quant = methods_by_name(spath, "AspectRatioGroupSampler", ["_compute_aspect_ratios", "_quantize"], {"bisect": bisect},
                                 nested_in="__init__")
    hw = [(480, 640)] * 13 + [(640, 480)] * 11 + [(500, 500)] * 3 + [(333, 500), (500, 333), (1, 3), (400, 401)]
    class DS:
        def __len__(self): return len(hw)
        def get_img_info(self, i): return {"height": hw[i][0], "width": hw[i][1]}
    out["sampler_hw"] = np.asarray(hw, np.int64)
    ratios = car(DS())
    out["sampler_ratios"] = np.asarray(ratios, np.float64)
    for b, bins in enumerate(([1], [0.5, 1, 2], [2, 0.75], [1.0, 1.0])):
        out[f"sampler_bins_{b}"] = np.asarray(bins, np.float64)
        out[f"sampler_groups_{b}"] = np.asarray(quant(ratios, bins), np.int64)
    out["sampler_bins_n"] = np.int32(4)
    runs = [(4, [1]), (3, [0.5, 1, 2]), (1, [1]), (5, [1])]
    out["sampler_runs_n"] = np.int32(len(runs))
    for r, (bs, bins) in enumerate(runs):
        gids = quant(ratios, bins)
        perms = [rng.permutation(len(hw)) for _ in range(3)]
        state = types.SimpleNamespace(world_size=1, group_ids=gids, batch_size=bs, buffer_per_group={k: [] for k in np.unique(gids).tolist()})
        out[f"sampler_run_{r}_batch_size"], out[f"sampler_run_{r}_bins"] = np.int32(bs), np.asarray(bins, np.float64)
        for e, perm in enumerate(perms):
            state.sample = lambda perm=perm: perm.tolist()
            got = [list(map(int, b)) for b in batch(state)]
            out[f"sampler_run_{r}_perm_{e}"] = np.asarray(perm, np.int64)
            out[f"sampler_run_{r}_flat_{e}"] = np.asarray([i for b in got for i in b], np.int64)
            out[f"sampler_run_{r}_count_{e}"] = np.int32(len(got))
    np.savez_compressed(os.path.join(HERE, "reference_batch_contract.npz"), **out)


def nms_fixtures():
    """Keep lists of the reference's own numpy NMS (layers/common/post_processing.py:106-132 py_cpu_nms) on seeded random boxes with
    DISTINCT scores (its argsort()[::-1] is not stable, so ties have no reference answer)."""
    (py_cpu_nms,) = functions_by_name(os.path.join(REF, "basedet/layers/common/post_processing.py"), ["py_cpu_nms"], {"np": np})
    rng = np.random.default_rng(2024)
    out = {}
    cases = []
    for i, (n, thr, spread) in enumerate([(1, 0.5, 300), (7, 0.5, 60), (64, 0.3, 200), (65, 0.7, 200), (200, 0.5, 300), (1000, 0.5, 600),
                                          (1000, 0.6, 250), (3000, 0.5, 900)]):
        xy = rng.uniform(0, spread, (n, 2)); wh = rng.uniform(4, 150, (n, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        if n >= 64:
            boxes[n // 2] = boxes[3]                                   # an exact duplicate (IoU 1) and a zero-area box
            boxes[n // 2 + 1, 2] = boxes[n // 2 + 1, 0]
        scores = rng.permutation(n).astype(np.float32) / np.float32(n) + np.float32(0.001)     # distinct
        dets = np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)
        keep = np.asarray(py_cpu_nms(dets, thr), np.int32)
        out[f"boxes_{i}"], out[f"scores_{i}"], out[f"thr_{i}"], out[f"keep_{i}"] = boxes, scores, np.float32(thr), keep
        cases.append(i)
    out["cases"] = np.asarray(cases, np.int32)
    np.savez(os.path.join(HERE, "reference_nms.npz"), **out)


def checkpoint_match_fixtures():
    """Outputs of the reference's name / shape matching (utils/checkpoint.py:13-29 get_name_matched_keys, get_shape_matched_keys;
    :40-90 full_match + _filter_unmatched_keys) on key tables shaped like the real use: an ImageNet ResNet file (no prefix, BatchNorm
    vectors dumped as (1, C, 1, 1)) into a RetinaNet state dict, a full detector checkpoint, ambiguous suffixes resolved by element
    count, unused keys.  Stored as JSON (names and shapes only)."""
    import json
    from typing import Dict, Set, Tuple
    names = ["get_name_matched_keys", "get_shape_matched_keys", "_filter_unmatched_keys", "full_match"]
    gn, gs, _, fm = functions_by_name(os.path.join(REF, "basedet/utils/checkpoint.py"), names,
                                      {"np": np, "Dict": Dict, "Set": Set, "Tuple": Tuple})

    class Shaped:                       # full_match only reads .shape of the checkpoint values
        def __init__(self, shape):
            self.shape = tuple(shape)

    model = {}
    def conv(n, co, ci, k): model[n + ".weight"] = (co, ci, k, k)
    def bn(n, c):
        for t in ("weight", "bias", "running_mean", "running_var"):
            model[f"{n}.{t}"] = (c,)
    conv("backbone.bottom_up.conv1", 64, 3, 7); bn("backbone.bottom_up.bn1", 64)
    for blk, (ci, ch, co) in enumerate([(64, 64, 256), (256, 64, 256)]):
        pre = f"backbone.bottom_up.layer1.{blk}"
        conv(pre + ".conv1", ch, ci, 1); bn(pre + ".bn1", ch); conv(pre + ".conv2", ch, ch, 3); bn(pre + ".bn2", ch)
        conv(pre + ".conv3", co, ch, 1); bn(pre + ".bn3", co)
        if blk == 0:
            conv(pre + ".downsample.0", co, ci, 1); bn(pre + ".downsample.1", co)
    for s, ci in ((3, 512), (4, 1024), (5, 2048)):
        conv(f"backbone.fpn_lateral{s}", 256, ci, 1); model[f"backbone.fpn_lateral{s}.bias"] = (256,)
        conv(f"backbone.fpn_output{s}", 256, 256, 3); model[f"backbone.fpn_output{s}.bias"] = (256,)
    for t in ("cls_subnet", "bbox_subnet"):
        for i in (0, 2, 4, 6):
            conv(f"head.{t}.{i}", 256, 256, 3); model[f"head.{t}.{i}.bias"] = (256,)
    conv("head.cls_score", 720, 256, 3); model["head.cls_score.bias"] = (720,)
    conv("head.bbox_pred", 36, 256, 3); model["head.bbox_pred.bias"] = (36,)

    cases = {}
    # (a) ImageNet backbone file: bare names, BN vectors (1, C, 1, 1), an fc layer nobody wants
    ck = {}
    for k, v in model.items():
        if k.startswith("backbone.bottom_up."):
            kk = k[len("backbone.bottom_up."):]
            ck[kk] = (1, v[0], 1, 1) if (len(v) == 1) else v
    ck["fc.weight"] = (1000, 2048); ck["fc.bias"] = (1000,)
    cases["imagenet_backbone"] = ck
    # (b) a full detector checkpoint: exact names
    cases["full_detector"] = dict(model)
    # (c) ambiguous suffixes: "0.weight" / "0.bias" hit both towers (same element counts -> must raise), "cls_score.weight" is unique,
    #     "conv2.weight" hits two blocks with equal shapes -> assert; "downsample.0.weight" unique
    cases["unique_suffixes"] = {"cls_score.weight": (720, 256, 3, 3), "downsample.0.weight": (256, 64, 1, 1), "bbox_pred.bias": (36,),
                                "fpn_lateral4.weight": (256, 1024, 1, 1), "not.there": (3,)}
    cases["ambiguous_by_count"] = {"conv1.weight": (64, 3, 7, 7)}              # stem (9408) vs layer1.0.conv1 (4096) vs layer1.1.conv1 (16384): count decides
    cases["ambiguous_assert"] = {"conv2.weight": (64, 64, 3, 3)}                # two blocks, same count -> AssertionError
    out = {"model": [[k, list(v)] for k, v in model.items()], "cases": {}}
    for name, ck in cases.items():
        rec = {"weights": [[k, list(v)] for k, v in ck.items()]}            # a LIST: the iteration order of the checkpoint matters
        try:
            mapping, unused = fm({k: Shaped(v) for k, v in ck.items()}, dict(model))
            rec["mapping"] = {k: v for k, v in mapping.items() if k is not None}
            rec["unused"] = list(unused)
        except AssertionError as e:
            rec["raises"] = "AssertionError"
        out["cases"][name] = rec
    out["name_matched"] = {q: sorted(gn(q, model.keys())) for q in ("conv1.weight", "bn1.weight", "weight", "head.cls_score.bias", "layer1.0.conv3.weight", "x")}
    out["shape_matched"] = {"(1,64,1,1)": sorted(gs((1, 64, 1, 1), {k: v for k, v in model.items() if k.endswith("bn1.weight")})),
                            "(256,)": sorted(gs((256,), {k: v for k, v in model.items() if k.endswith(".bias") and "head" in k}))}
    with open(os.path.join(HERE, "reference_checkpoint_match.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    nms_fixtures()
    checkpoint_match_fixtures()
    batch_contract_fixtures()
    # ---- 1. DummyLoader (basedet/utils/dummy.py:8-63) --------------------------------------
    dummy = load_by_path("ref_dummy", os.path.join(REF, "basedet/utils/dummy.py"))
    out = {}
    for tag, size in (("800x1344", (800, 1344)), ("512x512", (512, 512))):
        dl = dummy.DummyLoader(batch_size=2, output_size=size)
        out[f"anno_{tag}"] = dl.anno.astype(np.float32)
        out[f"im_info_{tag}"] = dl.im_info.astype(np.float32)
    # tiling rule for batch 5 (repeat 2, remainder 1) -- dummy.py:51-57 uses a float repeat count,
    # which numpy >= 2 rejects, so restate it with the integer quotient and record the result.
    dl = dummy.DummyLoader(batch_size=2, output_size=(800, 1344))
    b = next(dl)
    out["batch2_gt_boxes"] = b["gt_boxes"].astype(np.float32)
    out["batch2_im_info"] = b["im_info"].astype(np.float32)
    out["batch2_data_shape"] = np.asarray(b["data"].shape, np.int64)
    np.savez(os.path.join(HERE, "dummy_loader.npz"), **out)

    # ---- 2. reference unit-test known answers ------------------------------------------------
    kat = {}
    # tests/structures/test_boxes.py:15-34 (inputs), :38-46 iou, :48-57 ioa, :59-70 intersection,
    # :72-74 scale, :76-86 centers
    kat["boxes1"] = np.array([[0, 0, 1, 1], [0, 0, 1, 1]], np.float32)
    kat["boxes2"] = np.array([[0, 0, 1, 1], [0, 0, .5, 1], [0, 0, 1, .5], [0, 0, .5, .5], [.5, .5, 1, 1], [.5, .5, 1.5, 1.5]], np.float32)
    row = [1.0, 0.5, 0.5, 0.25, 0.25, 0.25 / (2 - 0.25)]
    kat["iou_1x2"] = np.array([row, row], np.float64)
    row = [1.0, 0.5, 0.5, 0.25, 0.25, 0.25]
    kat["ioa_2x1"] = np.array([row, row], np.float64).T
    kat["inter_1x2"] = np.array([row, row], np.float64)
    kat["centers_1"] = np.array([[0.5, 0.5], [0.5, 0.5]], np.float64)
    # tests/layers/test_postprocess.py:13-28
    kat["nms_boxes"] = np.array([[0, 0, 100, 100], [0, 0, 100.5, 100], [0, 0, 201, 200.5], [0, 0, 200.5, 200.5],
                                 [.5, .5, 100, 101], [.5, .5, 120.5, 120.5]], np.float32)
    kat["nms_scores"] = np.array([0.9, 0.8, 0.3, 0.7, 0.6, 0.4], np.float32)
    kat["nms_labels"] = np.array([1, 1, 1, 2, 2, 2], np.int32)
    kat["nms_iou_thresh"] = np.float32(0.4)
    kat["nms_keep"] = np.array([0, 3, 4, 2], np.int32)
    # tests/layers/test_preprocess.py:13-35 (shapes; and sum preserved for all-ones input)
    kat["pad_in_shapes"] = np.array([[1, 1, 1, 790, 790], [1, 1, 1, 799, 799], [1, 1, 1, 800, 800], [1, 1, 1, 801, 801],
                                     [1, 2, 10, 630, 630], [2, 2, 4, 639, 639]], np.int64)  # left-padded with 1s to rank 5
    kat["pad_in_rank"] = np.array([3, 3, 3, 3, 4, 5], np.int64)
    kat["pad_out_hw"] = np.array([[800, 800], [800, 800], [800, 800], [832, 832], [640, 640], [640, 640]], np.int64)
    # tests/layers/test_roi_pool.py:17-61: 5x5 arange feature, roi [0,1,1,3,3], 4x4 output
    kat["roi_feat"] = np.arange(25, dtype=np.float32).reshape(1, 1, 5, 5)
    kat["roi_rois"] = np.array([[0, 1, 1, 3, 3]], np.float32)
    kat["roi_align_4x4"] = np.array([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0], [12.0, 12.5, 13.0, 13.5]], np.float64)
    kat["roi_pool_4x4"] = np.array([[6, 7, 8, 8], [11, 12, 13, 13], [16, 17, 18, 18], [16, 17, 18, 18]], np.float64)
    np.savez(os.path.join(HERE, "reference_kat.npz"), **kat)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    sys.exit(main())
