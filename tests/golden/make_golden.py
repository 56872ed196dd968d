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


def main():
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
