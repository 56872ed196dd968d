"""Host-side pieces next to the inference / data path: the COCO box-AP restatement (hand-computable cases) and the
augmentation transforms (configs/detection_cfg.py:42-53).  No GPU."""
import numpy as np
import pytest

from basedet_amd.data import (Compose, RandomHorizontalFlip, ShortestEdgeResize, TestTimeCompose, ToMode, build_transform)
from basedet_amd.evaluators import COCOEvaluator, bbox_eval


def _gt(i, img, cat, box, crowd=0):
    return {"id": i, "image_id": img, "category_id": cat, "bbox": list(box), "area": box[2] * box[3], "iscrowd": crowd}


def _dt(img, cat, box, score):
    return {"image_id": img, "category_id": cat, "bbox": list(box), "score": score}


def test_perfect_detections_give_ap_one():
    gts = [_gt(1, 1, 1, (10, 10, 50, 50)), _gt(2, 1, 2, (100, 100, 20, 20)), _gt(3, 2, 1, (0, 0, 200, 200))]
    dts = [_dt(g["image_id"], g["category_id"], g["bbox"], 0.9) for g in gts]
    st = bbox_eval(gts, dts)["stats"]
    assert st[0] == pytest.approx(1.0) and st[1] == pytest.approx(1.0) and st[8] == pytest.approx(1.0)
    assert st[3] == pytest.approx(1.0)      # small: the 20x20 box
    assert st[4] == pytest.approx(1.0)      # medium: 50x50
    assert st[5] == pytest.approx(1.0)      # large: 200x200


def test_tp_fp_tp_curve():
    # two ground truths; detections by score: TP, FP, TP  ->  precision envelope 1.0 up to recall 0.5, 2/3 up to recall 1.0
    gts = [_gt(1, 1, 1, (0, 0, 40, 40)), _gt(2, 1, 1, (100, 100, 40, 40))]
    dts = [_dt(1, 1, (0, 0, 40, 40), 0.9), _dt(1, 1, (300, 300, 40, 40), 0.8), _dt(1, 1, (100, 100, 40, 40), 0.7)]
    r = bbox_eval(gts, dts)
    expect = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
    assert r["stats"][1] == pytest.approx(expect, abs=1e-9)       # AP50
    assert r["stats"][0] == pytest.approx(expect, abs=1e-9)       # exact boxes: same at every IoU threshold
    assert r["stats"][6] == pytest.approx(0.5)                     # AR@1: only the top detection counts
    assert r["stats"][8] == pytest.approx(1.0)


def test_iou_threshold_sweep():
    # a detection with IoU 0.6 counts at thresholds 0.50, 0.55, 0.60 only -> AP = 3/10
    gts = [_gt(1, 1, 1, (0, 0, 100, 100))]
    dts = [_dt(1, 1, (0, 0, 100, 60), 0.9)]                        # IoU = 6000 / 10000
    st = bbox_eval(gts, dts)["stats"]
    assert st[1] == pytest.approx(1.0) and st[2] == pytest.approx(0.0) and st[0] == pytest.approx(0.3)


def test_crowd_matches_are_ignored():
    # the second detection lies inside a crowd region (IoU = inter / det area = 1): neither TP nor FP
    gts = [_gt(1, 1, 1, (0, 0, 40, 40)), _gt(2, 1, 1, (100, 100, 200, 200), crowd=1)]
    dts = [_dt(1, 1, (0, 0, 40, 40), 0.9), _dt(1, 1, (150, 150, 30, 30), 0.8), _dt(1, 1, (160, 160, 30, 30), 0.7)]
    st = bbox_eval(gts, dts)["stats"]
    assert st[0] == pytest.approx(1.0) and st[8] == pytest.approx(1.0)


def test_missing_category_and_area_cells_are_minus_one():
    gts = [_gt(1, 1, 1, (0, 0, 200, 200))]
    st = bbox_eval(gts, [_dt(1, 1, (0, 0, 200, 200), 0.5)])["stats"]
    assert st[3] == -1.0 and st[4] == -1.0 and st[5] == pytest.approx(1.0)
    # ground truth but no detections at all: AP 0 (recall 0), not -1
    none = bbox_eval(gts, [])["stats"]
    assert none[0] == 0.0 and none[8] == 0.0


def test_max_dets_truncation():
    gts = [_gt(i, 1, 1, (20 * i, 0, 10, 10)) for i in range(12)]
    dts = [_dt(1, 1, (20 * i, 0, 10, 10), 0.99 - 0.01 * i) for i in range(12)]
    st = bbox_eval(gts, dts)["stats"]
    assert st[6] == pytest.approx(1 / 12) and st[7] == pytest.approx(10 / 12) and st[8] == pytest.approx(1.0)


def test_evaluator_format_roundtrip(tmp_path):
    ev = COCOEvaluator()
    rec = ev.postprocess({"boxes": np.array([[10, 20, 60, 100.0]]), "box_scores": np.array([0.75]), "box_labels": np.array([3])}, 7)
    empty = ev.postprocess({"boxes": np.zeros((0, 4)), "box_scores": np.zeros(0), "box_labels": np.zeros(0)}, 8)
    res = ev.format([rec, empty])
    assert res == [{"image_id": 7, "bbox": [10.0, 20.0, 50.0, 80.0], "score": 0.75, "category_id": 4}]
    path = ev.save_results([rec, empty], str(tmp_path / "predict_coco.json"))
    ann = {"images": [{"id": 7}, {"id": 8}], "categories": [{"id": 4}],
           "annotations": [_gt(1, 7, 4, (10, 20, 50, 80))]}
    out = ev.evaluate(path, ann)
    assert out["AP"] == pytest.approx(1.0) and out["AR100"] == pytest.approx(1.0)


# ------------------------------------------------------------------------------------------------ transforms

def test_shortest_edge_resize_shapes_and_boxes():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, size=(480, 640, 3), dtype=np.uint8)
    t = ShortestEdgeResize(min_size=(800,), max_size=1333, sample_style="choice", rng=rng)
    boxes = np.array([[64, 48, 320, 240]], dtype=np.float32)
    t.order = ("image", "boxes", "boxes_category")
    out, b, c = t.apply((img, boxes, np.array([5])))
    assert out.shape == (800, 1067, 3) and out.dtype == np.uint8          # 640 * 800/480 = 1066.67 -> 1067
    np.testing.assert_allclose(b, boxes * np.array([1067 / 640, 800 / 480, 1067 / 640, 800 / 480]), rtol=1e-6)
    assert c[0] == 5
    # long edge capped at max_size
    t2 = ShortestEdgeResize(min_size=800, max_size=1333, sample_style="choice", rng=rng)
    out2 = t2.apply(rng.integers(0, 255, size=(300, 900, 3), dtype=np.uint8))
    assert out2.shape[:2] == (444, 1333)


def test_resize_constant_and_gradient_images():
    t = ShortestEdgeResize(min_size=64, max_size=1000, sample_style="choice")
    const = np.full((32, 48, 3), 77, dtype=np.uint8)
    assert (t.apply(const) == 77).all()
    ramp = np.tile(np.arange(48, dtype=np.float32)[None, :, None], (32, 1, 1))
    out = ShortestEdgeResize(min_size=64, max_size=1000, sample_style="choice").apply(ramp)
    mid = out[10, 4:-4, 0]
    np.testing.assert_allclose(np.diff(mid), 0.5, atol=1e-4)           # 2x upscale of a unit ramp: slope 0.5 away from the edges


def test_flip_and_compose_order():
    rng = np.random.default_rng(3)
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    boxes = np.array([[1, 0, 3, 2]], dtype=np.float32)
    pipe = Compose([RandomHorizontalFlip(prob=1.0, rng=rng), ToMode("CHW")])
    o, b, c = pipe((img, boxes, np.array([1])))
    assert o.shape == (3, 4, 6) and (o[:, :, 0] == img[:, 5].T).all()
    np.testing.assert_array_equal(b, [[3, 0, 5, 2]])
    o2, b2, _ = Compose([RandomHorizontalFlip(prob=0.0, rng=rng)])((img, boxes, np.array([1])))
    assert (o2 == img).all() and (b2 == boxes).all()


def test_test_time_compose_im_info():
    pipe = build_transform(mode="test")
    assert isinstance(pipe, TestTimeCompose)
    img = np.zeros((500, 375, 3), dtype=np.uint8)
    out, info = pipe(img)
    assert out.shape == (1, 3, 1067, 800) and out.dtype == np.float32
    np.testing.assert_array_equal(info, [[1067, 800, 500, 375]])
    train = build_transform(mode="train", rng=np.random.default_rng(1))
    o, b, c = train((img, np.array([[10, 10, 100, 100]], dtype=np.float32), np.array([2])))
    assert o.shape[0] == 3 and min(o.shape[1:]) in (640, 672, 704, 736, 768, 800)


def _textbook_ap(gts, dts, thr):
    """An independent restatement of box AP at one IoU threshold for ONE category without crowds / area ranges, written from the
    definition (not from bbox_eval's code): detections of all images by descending score; each takes the unmatched ground truth of
    its image with the highest IoU >= thr; precision envelope sampled at recall 0, 0.01, ..., 1."""
    def iou(a, b):
        ax2, ay2, bx2, by2 = a[0] + a[2], a[1] + a[3], b[0] + b[2], b[1] + b[3]
        iw, ih = min(ax2, bx2) - max(a[0], b[0]), min(ay2, by2) - max(a[1], b[1])
        inter = max(iw, 0.0) * max(ih, 0.0)
        return inter / (a[2] * a[3] + b[2] * b[3] - inter)
    order = sorted(range(len(dts)), key=lambda i: (-dts[i]["score"], i))
    used = set()
    tp = []
    for i in order:
        d = dts[i]
        best, arg = thr, None
        for j, g in enumerate(gts):
            if g["image_id"] != d["image_id"] or j in used:
                continue
            v = iou(d["bbox"], g["bbox"])
            if v >= best:
                best, arg = v, j
        if arg is not None:
            used.add(arg)
        tp.append(arg is not None)
    tp = np.asarray(tp, dtype=np.float64)
    ctp, cfp = np.cumsum(tp), np.cumsum(1 - tp)
    rec, prec = ctp / len(gts), ctp / np.maximum(ctp + cfp, 1e-300)
    ap = 0.0
    for r in np.linspace(0, 1, 101):
        m = prec[rec >= r - 1e-12]
        ap += (m.max() if m.size else 0.0) / 101
    return ap


def test_bbox_eval_agrees_with_an_independent_textbook_ap():
    """Random detection sets (several images, one category, jittered copies of the ground truth + clutter, distinct scores): the AP at
    every IoU threshold of bbox_eval equals an AP written independently from the definition -- pycocotools itself is not available here,
    so this is the strongest pin the evaluator can get in this container."""
    rng = np.random.default_rng(12)
    for trial in range(6):
        gts, dts, k = [], [], 0
        for img in range(1, 5):
            for _ in range(rng.integers(1, 6)):
                x, y, w, h = rng.uniform(0, 300), rng.uniform(0, 300), rng.uniform(120, 200), rng.uniform(120, 200)   # all "large": one area cell
                k += 1
                gts.append(_gt(k, img, 1, (x, y, w, h)))
                for _ in range(rng.integers(0, 3)):
                    j = rng.normal(0, 12, 4)
                    dts.append(_dt(img, 1, (x + j[0], y + j[1], max(w + j[2], 5), max(h + j[3], 5)), 0.0))
            for _ in range(rng.integers(0, 4)):
                dts.append(_dt(img, 1, (rng.uniform(0, 400), rng.uniform(0, 400), rng.uniform(100, 180), rng.uniform(100, 180)), 0.0))
        for d, s in zip(dts, rng.permutation(len(dts))):
            d["score"] = 0.05 + 0.9 * (s + 1) / (len(dts) + 1)
        r = bbox_eval(gts, dts)
        P = r["precision"]                                   # (T, R, K, A, M)
        for ti, thr in enumerate(np.linspace(0.5, 0.95, 10)):
            want = _textbook_ap(gts, dts, thr)
            got = float(P[ti, :, 0, 0, 2].mean())
            assert got == pytest.approx(want, abs=1e-9), (trial, thr, got, want)
        assert r["stats"][0] == pytest.approx(np.mean([_textbook_ap(gts, dts, t) for t in np.linspace(0.5, 0.95, 10)]), abs=1e-9)


# ---- hand-computed cells for the cases pycocotools' protocol is known for ------------------------------------
# Every expected number below is worked out in the comment from COCOeval's published rules (evaluateImg / accumulate), not read off the
# implementation: a change of the matching or interpolation rule moves them.

def test_crowd_region_is_rematched_and_uses_detection_area_iou():
    """g1 normal, g2 crowd [20, 40) x [0, 10).  By score: d1, d2 lie inside the crowd (plain IoU 100 / 200 = 0.5 would fail every threshold
    above 0.50; the crowd rule inter / det-area gives 1.0), d3 is g1 exactly.  d1 matches the crowd -> ignored; d2 matches the SAME crowd
    again -> ignored (matched once only it would be a false positive in front of the true positive: precision 1/2); d3 is the one true
    positive of one countable ground truth: tp = [0, 0, 1], fp = [0, 0, 0] -> precision 1 at every recall point and IoU threshold."""
    gts = [_gt(1, 1, 1, (0, 0, 10, 10)), _gt(2, 1, 1, (20, 0, 20, 10), crowd=1)]
    dts = [_dt(1, 1, (20, 0, 10, 10), 0.9), _dt(1, 1, (30, 0, 10, 10), 0.8), _dt(1, 1, (0, 0, 10, 10), 0.7)]
    r = bbox_eval(gts, dts)
    assert np.all(r["precision"][:, :, 0, 0, 2] == pytest.approx(1.0))
    assert r["stats"][0] == pytest.approx(1.0) and r["stats"][2] == pytest.approx(1.0) and r["stats"][8] == pytest.approx(1.0)
    # the same detections against a NON-crowd g2: d1 takes it at IoU 0.5 (threshold 0.50 only), d2 is a false positive:
    #   t = 0.50: npig 2, by score tp [1, 1, 2], fp [0, 1, 1] -> rc [.5, .5, 1], pr [1, .5, 2/3] -> envelope [1, 2/3, 2/3]:
    #             51 recall points (<= .5) at 1, 50 at 2/3
    #   t > 0.50: d1, d2 false positives: tp [0, 0, 1], fp [1, 2, 2] -> rc [0, 0, .5], pr 1/3 up to recall .5 (51 points), 0 beyond
    gts[1]["iscrowd"] = 0
    P = bbox_eval(gts, dts)["precision"][:, :, 0, 0, 2]
    assert P[0].mean() == pytest.approx((51 + 50 * 2 / 3) / 101, abs=1e-12)
    for ti in range(1, 10):
        assert P[ti].mean() == pytest.approx(51 / 3 / 101, abs=1e-12)


def test_area_range_ignores_ground_truth_and_unmatched_detections():
    """g1 20 x 20 (area 400: small), g2 50 x 50 (2 500: medium).  By score: d3 (30 x 30 = 900, matches nothing), d1 = g1, d2 = g2,
    d4 (50 x 50, matches nothing).
      all:    FP TP TP FP -> tp [0,1,2,2], fp [1,1,1,2], rc [0,.5,1,1], pr [0,1/2,2/3,1/2] -> envelope 2/3 everywhere: AP 2/3
      small:  g2 ignored; d3 FP (in range), d1 TP, d2 matches the ignored g2 -> ignored, d4 unmatched and out of range -> ignored:
              tp [0,1,1,1], fp [1,1,1,1], one countable gt -> pr envelope 1/2 at every recall point: AP 1/2
      medium: g1 ignored; d3 unmatched, out of range -> ignored; d1 matches the ignored g1 -> ignored; d2 TP; d4 FP (in range):
              tp [0,0,1,1], fp [0,0,0,1] -> pr [0,0,1,1/2] -> envelope [1,1,1,1/2]; recall reaches 1 at index 2: AP 1
      large:  no ground truth in range: -1"""
    gts = [_gt(1, 1, 1, (0, 0, 20, 20)), _gt(2, 1, 1, (100, 100, 50, 50))]
    dts = [_dt(1, 1, (300, 300, 30, 30), 0.95), _dt(1, 1, (0, 0, 20, 20), 0.9), _dt(1, 1, (100, 100, 50, 50), 0.8),
           _dt(1, 1, (400, 400, 50, 50), 0.6)]
    st = bbox_eval(gts, dts)["stats"]
    assert st[0] == pytest.approx(2 / 3, abs=1e-12)
    assert st[3] == pytest.approx(0.5, abs=1e-12)
    assert st[4] == pytest.approx(1.0, abs=1e-12)
    assert st[5] == -1.0
    assert st[9] == pytest.approx(1.0) and st[10] == pytest.approx(1.0) and st[11] == -1.0          # ARs, ARm, ARl


def test_max_dets_is_applied_per_image_before_pooling():
    """Image 1: d1 (0.9, matches nothing), d2 (0.8) = g1.  Image 2: d3 (0.7) = g2.
      maxDets 1:   image 1 keeps d1 only (its true positive is cut off PER IMAGE, although d2 outscores image 2's d3): pooled FP, TP ->
                   tp [0,1], fp [1,1], npig 2 -> rc [0,.5], pr envelope [.5,.5]: 51 recall points at 1/2, recall .5
      maxDets 100: FP TP TP -> tp [0,1,2], fp [1,1,1] -> pr [0,1/2,2/3] -> envelope 2/3 at all 101 points, recall 1"""
    gts = [_gt(1, 1, 1, (0, 0, 40, 40)), _gt(2, 2, 1, (0, 0, 40, 40))]
    dts = [_dt(1, 1, (200, 200, 40, 40), 0.9), _dt(1, 1, (0, 0, 40, 40), 0.8), _dt(2, 1, (0, 0, 40, 40), 0.7)]
    r = bbox_eval(gts, dts)
    assert r["precision"][0, :, 0, 0, 0].mean() == pytest.approx(51 * 0.5 / 101, abs=1e-12)
    assert r["recall"][0, 0, 0, 0] == pytest.approx(0.5) and r["stats"][6] == pytest.approx(0.5)
    assert r["stats"][0] == pytest.approx(2 / 3, abs=1e-12)
    assert r["stats"][7] == pytest.approx(1.0) and r["stats"][8] == pytest.approx(1.0)


def test_score_ties_keep_image_order_and_iou_equal_to_threshold_matches():
    """Pooled detections are sorted by score with a STABLE sort: equal scores keep the order of the image ids.  Image 1 holds a true
    positive, image 2 a false positive and an unmatched ground truth, both detections at score 0.5:
      image 1 first: tp [1,1], fp [0,1], npig 2 -> rc [.5,.5], pr [1,.5]: recall points 0 ... 0.50 (51 of 101) read precision 1 -> 51/101
      ids swapped:   tp [0,1], fp [1,1] -> pr envelope [.5,.5] -> 25.5/101
    recThrs[50] must compare equal to the recall 1/2 (searchsorted side = left)."""
    def case(tp_img, fp_img):
        gts = [_gt(1, tp_img, 1, (0, 0, 40, 40)), _gt(2, fp_img, 1, (0, 0, 40, 40))]
        dts = [_dt(tp_img, 1, (0, 0, 40, 40), 0.5), _dt(fp_img, 1, (200, 200, 40, 40), 0.5)]
        return bbox_eval(gts, dts)["stats"][0]
    assert case(1, 2) == pytest.approx(51 / 101, abs=1e-12)
    assert case(2, 1) == pytest.approx(25.5 / 101, abs=1e-12)
    # IoU exactly 0.5 (100 / 200) matches at the 0.50 threshold (the rule is "iou < threshold: no match") and at no other: AP = 1 / 10
    st = bbox_eval([_gt(1, 1, 1, (0, 0, 10, 10))], [_dt(1, 1, (0, 0, 10, 20), 0.9)])["stats"]
    assert st[1] == pytest.approx(1.0) and st[2] == pytest.approx(0.0) and st[0] == pytest.approx(0.1)
