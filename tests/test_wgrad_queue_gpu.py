"""The QUEUED weight-gradient path of a training step (round 4's default; round 5: opt-in, WGRAD_QUEUE = "bucket" or a byte threshold --
"layer" measured 0.6 % faster): partial sums of a whole gradient bucket queued in one arena, ONE fixed-order reduce per bucket
(models/fpn_base.py).  The first backward pass of a model runs un-queued (it sizes the arena),
so a test that builds a model and calls backward() once never reaches the queued path: here every model runs forward + backward TWICE
and a third time after a forced overflow, against WGRAD_QUEUE = "layer" (one reduce per layer, rounds 1-3) and an integer byte
threshold -- every gradient of the parameter arena bit for bit (replaces autodiff: basedet/solver/default_solver.py:118-124)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZE = (320, 416)


def _batch(N, size=SIZE):
    from basedet_amd.utils import DummyLoader
    b = next(DummyLoader(N, size, seed=0))
    return {"data": torch.from_numpy((b["data"] * 255).astype(np.float32)).cuda(), "gt_boxes": torch.from_numpy(b["gt_boxes"]).cuda(),
            "im_info": torch.from_numpy(b["im_info"]).cuda()}


def _build(kind, mode):
    from basedet_amd.configs import FasterRCNNConfig, FCOSConfig, RetinaNetConfig
    from basedet_amd.models import FCOS, FasterRCNN, RetinaNet, params as P
    cfg = {"retinanet": RetinaNetConfig, "fcos": FCOSConfig, "faster_rcnn": FasterRCNNConfig}[kind]()
    cfg.MODEL.BATCHSIZE = 2
    cfg.MODEL.WGRAD_QUEUE = mode
    if kind == "retinanet":
        return RetinaNet(cfg, params=P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2))
    if kind == "fcos":
        p = P.init_fcos_params(cfg, seed=0, residual_gamma=0.2)
        p["head.bbox_pred.bias"] = np.full_like(p["head.bbox_pred.bias"], 0.5)
        return FCOS(cfg, params=p)
    return FasterRCNN(cfg, params=P.init_faster_rcnn_params(cfg, 0, residual_gamma=0.2))


def _keys(model, batch, kind):
    if kind != "faster_rcnn":
        return batch
    pl = model._plan(2, SIZE[0], SIZE[1])
    rng = np.random.default_rng(5)
    G = batch["gt_boxes"].shape[1]
    keys = dict(rpn_pos=rng.random((2, pl.A_total), dtype=np.float32), rpn_neg=rng.random((2, pl.A_total), dtype=np.float32),
                rcnn_fg=rng.random((2, pl.rois.shape[1] + G), dtype=np.float32), rcnn_bg=rng.random((2, pl.rois.shape[1] + G), dtype=np.float32))
    return dict(batch, sample_keys=keys)


def _grads(model, batch, on_bucket_ready=None):
    model(batch)
    model.backward(on_bucket_ready)
    torch.cuda.synchronize()
    return model.arena.g.clone()


@pytest.mark.parametrize("kind", ["retinanet", "fcos", "faster_rcnn"])
def test_queued_weight_gradients_equal_the_per_layer_ones_bit_for_bit(kind):
    batch = _batch(2)
    ref_model = _build(kind, "layer")
    b = _keys(ref_model, batch, kind)
    ref = _grads(ref_model, b)
    assert float(ref.abs().max()) > 0
    # (round 5: RoIAlign's backward is the tiled fixed-order sum -- Faster R-CNN is bitwise reproducible like the other two)
    def same(a, tag):
        assert torch.equal(a, ref), tag

    for mode in ("bucket", 48 << 20):
        m = _build(kind, mode)
        first = _grads(m, b)                  # the sizing pass: un-queued
        assert m._wq_arena is not None and m._wq_peak > 0
        peak = m._wq_peak
        second = _grads(m, b)                 # queued: every layer in the arena, one reduce per bucket (or per 48 MB)
        assert m._wq is not None and m._wq.pending() == 0
        same(first, (mode, "first")); same(second, (mode, "second"))
        assert torch.equal(first, second)
        # the arena holds the LARGEST flush interval, not the sum over the pass
        total = sum(m._wq_need.values())
        assert peak < total, (peak, total)
        # a forced overflow (arena cut to a quarter): the layers that do not fit run un-queued -- same bits -- and the arena is re-grown
        m._wq_arena = m._wq_arena[: m._wq_arena.numel() // 4].clone()
        third = _grads(m, b)
        same(third, (mode, "overflow"))
        assert m._wq_arena.numel() * 4 >= peak
    # the gradient buckets are closed in backward order with the side stream's work flushed: what the all-reduce hook sees at
    # "bucket ready" time is final (a forced hook run: the hook copies each bucket as it is announced)
    m = _build(kind, "bucket")
    _grads(m, b)
    from basedet_amd.solver import DetSolver
    solver = DetSolver.build(m.cfg, m)
    seen = {}

    def hook(name, side):
        for s in side:
            torch.cuda.current_stream().wait_stream(s)
        lo, hi = solver.buckets.ranges[name]
        seen[name] = m.arena.g[lo:hi].clone()

    final = _grads(m, b, hook)
    assert set(seen) == set(solver.buckets.ranges), (sorted(seen), sorted(solver.buckets.ranges))
    for name, (lo, hi) in solver.buckets.ranges.items():
        assert torch.equal(seen[name], final[lo:hi]), name


def test_retinanet_head_towers_on_two_streams_equal_the_serial_head_bit_for_bit():
    """MODEL.HEAD_TOWERS_CONCURRENT (round 6, default on): the box tower's forward launches on the model's side stream beside the class tower's.
    Two steps each (the second one runs behind a backward pass that used the same side stream): logits, box offsets and every gradient of
    the parameter arena must be the same bits as with the towers one after the other."""
    from basedet_amd.configs import RetinaNetConfig
    from basedet_amd.models import RetinaNet, params as P
    batch = _batch(2)
    out = {}
    for conc in (False, True):
        cfg = RetinaNetConfig()
        cfg.MODEL.BATCHSIZE = 2
        cfg.MODEL.HEAD_TOWERS_CONCURRENT = conc
        m = RetinaNet(cfg, params=P.init_retinanet_params(cfg, seed=0, residual_gamma=0.2))
        assert m.async_wgrad and m._tstream is not None
        res = []
        for _ in range(2):
            losses = m(batch)
            m.backward()
            torch.cuda.synchronize()
            res.append((float(losses["total_loss"]), m._cur.logits.clone(), m._cur.offsets.clone(), m.arena.g.clone()))
        out[conc] = res
    for a, b in zip(out[False], out[True]):
        assert abs(a[0] - b[0]) <= 1e-5 * abs(a[0])          # (the reported loss scalars are atomic sums: equal to rounding, run to run)
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)
    assert float(out[True][0][3].abs().max()) > 0
