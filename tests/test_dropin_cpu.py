"""Drop-in check of the `basedet` import alias (CPU; no kernels run).

1. In the build container only (skipped where /root/reference is absent, e.g. on the GPU box): the reference's own
   playground/examples/retinanet/res50_coco_800size_1x/config.py is imported BY PATH against the alias -- `from basedet.configs import
   RetinaNetConfig` resolves to this repo -- and its `Cfg()` instantiates; the names the training entry resolves from it
   (registers.models / registers.solvers / registers.trainers, tools/det_train.py:111, configs/detection_cfg.py:55-63) exist.
   `megfile` (a path-joining helper the config imports; not installed here) is stood in by os.path.join inside the child process the
   config is executed in.
2. Everywhere: the alias exposes the operator surface under the reference's module paths and names."""
import importlib.util
import os
import sys
import types

import pytest

REF_CFG = "/root/reference/playground/examples/retinanet/res50_coco_800size_1x/config.py"


def _load_by_path(path):
    spec = importlib.util.spec_from_file_location("playground_cfg_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_CHILD = r"""
import importlib.util, json, os, sys, types
root, path, rel = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
import basedet  # noqa: F401  (the alias package at the repo root)
if "megfile" not in sys.modules:          # a path-joining helper the configs import; not installed here
    stub = types.ModuleType("megfile")
    stub.smart_path_join = lambda *a: os.path.join(*a)
    sys.modules["megfile"] = stub
spec = importlib.util.spec_from_file_location("playground_cfg_under_test", path)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
cfg = mod.Cfg()
from basedet.configs import DetectionConfig
from basedet.utils import registers
import basedet.models, basedet.solver, basedet.engine  # noqa: F401,E401
model_cls = registers.models.get(cfg.MODEL.NAME)
solver_builder = registers.solvers.get(cfg.SOLVER.BUILDER_NAME)
trainer_cls = registers.trainers.get(cfg.TRAINER.NAME)
print("RESULT " + json.dumps(dict(
    is_cfg=isinstance(cfg, DetectionConfig), out_dir=cfg.GLOBAL.OUTPUT_DIR,
    resolved=bool(callable(model_cls) and hasattr(solver_builder, "build") and hasattr(trainer_cls, "train")),
    builders=all(hasattr(cfg, n) for n in ("build_model", "build_solver", "build_trainer")),
    reduce=cfg.SOLVER.REDUCE_MODE, freeze=cfg.MODEL.BACKBONE.FREEZE_AT, classes=cfg.DATA.NUM_CLASSES)))
"""


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("rel", ["retinanet/res50_coco_800size_1x", "fcos", "faster_rcnn/res50_coco_800size_1x", "atss", "freeanchor", "ota"])
def test_reference_playground_config_loads_against_the_alias(rel):
    """The reference's config file is untrusted content: it is executed in a CHILD process (no GPU, nothing of this test process's state),
    which reports what the training entry would resolve from it."""
    import json
    import subprocess
    path = os.path.join("/root/reference/playground/examples", rel, "config.py")
    if not os.path.exists(path):
        pytest.skip(path + " absent")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CHILD, root, path, rel], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-1000:]
    res = json.loads(line[-1][7:])
    assert res["is_cfg"] and res["resolved"] and res["builders"]
    assert res["out_dir"].endswith(os.path.join("examples", rel))
    # the values the hot path reads are the reference's (configs/det_model/*.py, configs/extra_cfg.py)
    assert res["reduce"] == "MEAN" and res["freeze"] == 2 and res["classes"] == 80


def test_alias_exposes_the_operator_surface():
    import basedet
    from basedet import layers, structures
    import basedet_amd
    assert basedet.layers is basedet_amd.layers and sys.modules["basedet.configs"] is basedet_amd.configs
    for name in ("sigmoid_focal_loss", "smooth_l1_loss", "iou_loss", "binary_cross_entropy", "weighted_cross_entropy", "box_iou",
                 "roi_pool", "assign_rois", "sample_labels", "Matcher", "batched_nms", "post_processing", "data_to_input",
                 "get_padded_tensor", "permute_to_N_Any_K", "DefaultAnchorGenerator", "AnchorPointGenerator", "FPN", "RetinaNetHead",
                 "PointHead", "resnet50", "build_backbone"):
        assert callable(getattr(layers, name)), name
    for name in ("Boxes", "BoxCoder", "PointCoder", "Container", "box_iou", "box_ioa", "box_center"):
        assert hasattr(structures, name), name
    for m in ("filter_by_size", "cat", "scale", "clip", "iou", "ioa", "giou", "intersection", "__getitem__"):
        assert hasattr(structures.Boxes, m), m
    from basedet.tools.det_train import main, default_parser  # noqa: F401  (the basedet_train entry)


def test_matcher_keeps_the_reference_constructor_contract():
    from basedet.layers import Matcher
    thr = [0.4, 0.5]
    m = Matcher(thr, [0, -1, 1], allow_low_quality_matches=True)
    assert m.thresholds == [-float("inf"), 0.4, 0.5, float("inf")] and m.labels == [0, -1, 1]
    with pytest.raises(AssertionError):
        Matcher([0.5, 0.4], [0, -1, 1])
    with pytest.raises(AssertionError):
        Matcher([0.5], [0, -1, 1])


def test_pmc_traffic_parser_and_bench_fallback(tmp_path, monkeypatch):
    """scripts/pmc_traffic.load (what bench.py's in-run traffic measurement parses) on a synthetic rocprofv3 counter_collection.csv:
    kernel names are normalised (anonymous namespace, template arguments; the two BK instances of the generic kernel stay apart) and
    averaged per launch; bench.measure_traffic reports a reason instead of failing when it cannot run the profiler."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    pmc_traffic = importlib.import_module("pmc_traffic")
    d = tmp_path / "fetch" / "host" / "pid"
    d.mkdir(parents=True)
    rows = ["Kernel_Name,Counter_Name,Counter_Value",
            '"void (anonymous namespace)::conv1x1_dense_kernel<1, 4, false>((anonymous namespace)::P1)",FETCH_SIZE,100',
            '"void (anonymous namespace)::conv1x1_dense_kernel<1, 3, true>((anonymous namespace)::P1)",FETCH_SIZE,300',
            '"void (anonymous namespace)::conv_igemm_kernel<32, false, true>(igemm::IgemmParams)",FETCH_SIZE,50',
            '"void (anonymous namespace)::conv_igemm_kernel<64, false, true>(igemm::IgemmParams)",FETCH_SIZE,70',
            '"(anonymous namespace)::sgd_kernel(float*, float*, float const*, long long)",FETCH_SIZE,10',
            '"(anonymous namespace)::sgd_kernel(float*, float*, float const*, long long)",WRITE_SIZE,999']
    (d / "1_counter_collection.csv").write_text("\n".join(rows) + "\n")
    tot, cnt = pmc_traffic.load(str(tmp_path / "fetch"), "FETCH_SIZE")
    assert tot["conv1x1_dense_kernel"] == 400 and cnt["conv1x1_dense_kernel"] == 2
    assert tot["conv_igemm_kernel<32>"] == 50 and tot["conv_igemm_kernel<64>"] == 70
    assert tot["sgd_kernel"] == 10 and cnt["sgd_kernel"] == 1
    bench = importlib.import_module("bench")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "x")                     # "already under a profiler": skipped with a reason, no child process
    res, note = bench.measure_traffic(None)
    assert res == {} and "profiler" in note
