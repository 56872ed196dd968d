"""Full-resolution parity against the CPU oracle for BASELINE configs C2 / C3 / C4: ResNet-50 FPN, batch 2, 800 x 1344 (22 400
locations, 201 600 anchors per image; Faster R-CNN: 268 569 anchors, 2000 -> 1000 proposals, 512 RoIs per image -- the CONFIGURED
sizes, configs/det_model/faster_rcnn_cfg.py).  bench.py's CPU-baseline leg runs the same oracle step in ~20 s, so the comparison is
affordable; it sees what size-independent properties cannot: a tile that is consistently wrong at large grid indices.

Per model: discrete targets (labels, matched anchors, RPN labels, sampled RoIs) bit-exact; losses <= 2e-2 and logits rel-L2 <= 2e-2
against the plain fp32 oracle; EVERY parameter gradient against the plain fp32 oracle bounded in DIRECTION (per-parameter cosine and
whole-model cosine, norm ratio: bf16 forward differences flip ReLU gates near zero, so a rel-L2 bound against the plain oracle would
assert little -- but a wrong small parameter cannot hide in a global cosine), and within 2e-2 (Faster R-CNN: 1e-2) in rel-L2 against
the oracle evaluated on the HIP run's own stored activations (identical gates: the backward pass is then the same linear map on both
sides)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZE = (800, 1344)
N = 2
# Against the PLAIN fp32 oracle a parameter gradient differs by bf16 forward noise flipping ReLU gates near zero: not a tight bound (the
# tight one is the injected-activation comparison, 2e-2).  Round 5: the direction is bounded instead of a 0.40 rel-L2 that asserted
# almost nothing -- per-parameter cosine and the whole-model cosine; the rel-L2 maxima are printed as a diagnostic (-s).
MIN_COS_PARAM = 0.975       # RetinaNet-R50: observed worst parameter 0.9909 (layer2.1.conv1), whole model 0.99954, norm ratio 1.003
MIN_COS_MODEL = 0.998
FCOS_MIN_COS = (0.93, 0.99)  # FCOS-R50 (GroupNorm towers amplify the flipped gates): observed 0.9550 (layer2.3.conv1) / 0.99659
# Faster R-CNN against the PLAIN oracle (whose fp32 scores may order proposals differently): observed worst parameter 0.99798
# (backbone.fpn_output4.weight), whole model 0.99989, norm ratio 0.996; against the injected oracle worst rel-L2 2.3e-3.
FRCNN_MIN_COS = (0.99, 0.999)


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def _check_grads(names, got, plain, injected, tag, min_cos_param=MIN_COS_PARAM, min_cos_model=MIN_COS_MODEL, inj_bound=2e-2):
    worst_plain, worst_inj, worst_cos = ("", 0.0), ("", 0.0), ("", 1.0)
    dot = na = nb = 0.0
    for n in names:
        g, p_ = got[n].double().reshape(-1).cpu(), plain[n].detach().double().reshape(-1)
        rp = _rel(g, p_)
        ri = _rel(got[n], injected[n].detach())
        c = _cos(g, p_)
        dot += float((g * p_).sum()); na += float((g * g).sum()); nb += float((p_ * p_).sum())
        if rp > worst_plain[1]:
            worst_plain = (n, rp)
        if ri > worst_inj[1]:
            worst_inj = (n, ri)
        if c < worst_cos[1]:
            worst_cos = (n, c)
    model_cos = dot / (np.sqrt(na) * np.sqrt(nb) + 1e-300)
    print(f"[{tag}] per-parameter gradient vs the plain oracle: worst rel-L2 {worst_plain} (diagnostic), worst cosine {worst_cos}, "
          f"whole-model cosine {model_cos:.5f}, norm ratio {np.sqrt(na / nb):.4f}; vs the injected oracle: worst rel-L2 {worst_inj}")
    assert worst_cos[1] > min_cos_param, worst_cos
    assert model_cos > min_cos_model, model_cos
    assert 0.9 < np.sqrt(na / nb) < 1.1, np.sqrt(na / nb)
    assert worst_inj[1] < inj_bound, worst_inj


def _check_forward_layers(orc, batch, fn, tag, bound=2e-2):
    """EVERY stored activation of the HIP forward (block mids / shortcuts / outputs, laterals, pyramid levels, tower activations, logits)
    against the plain fp32 oracle at full size, layer by layer: a forward error that the end-to-end logits bound would let through --
    one that flips ReLU gates in one layer only -- shows up here as that layer's rel-L2.  Observed: bf16 storage noise grows from ~2e-3
    after the stem to ~1e-2 at the deepest layers."""
    with torch.no_grad():
        getattr(orc, fn)(batch)
    rec = {k: v for k, v in orc.record.items() if k != "_compare"}
    assert len(rec) > 60, sorted(rec)[:5]
    worst = max(rec.items(), key=lambda kv: kv[1])
    print(f"[{tag}] per-layer forward rel-L2 vs the fp32 oracle over {len(rec)} stored activations: worst {worst}")
    assert worst[1] < bound, worst


def test_retinanet_r50_full_size_matches_oracle():
    from basedet_amd.models import RetinaNet, params as P
    from oracle.model import Oracle
    from tests.test_model_gpu import _setup
    cfg, params, batch = _setup("resnet50", N, SIZE)
    model = RetinaNet(cfg, params=params)
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    ref, aux = orc.retinanet_losses(batch)
    ref_grads = orc.grads(ref["total_loss"])
    out = model(batch)
    pl = model._cur
    assert pl.labels.shape == (N, 201600)
    assert np.array_equal(pl.labels.cpu().numpy(), aux["labels"])
    assert int(pl.num_fg.item()) == aux["num_fg"]
    for k in ("cls_loss", "reg_loss", "total_loss"):
        got, want = float(out[k]), float(ref[k].detach())
        assert abs(got - want) / abs(want) < 2e-2, (k, got, want)
    K = cfg.DATA.NUM_CLASSES
    assert _rel(pl.logits.float().cpu().view(-1, K), aux["logits"].detach()) < 2e-2
    model.backward()
    torch.cuda.synchronize()
    got = model.reference_grads()
    acts = model.debug_activations()
    orc2 = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=acts)
    l2, _ = orc2.retinanet_losses(batch)
    _check_grads(names, got, ref_grads, orc2.grads(l2["total_loss"]), "RetinaNet-R50 2x800x1344")
    _check_forward_layers(Oracle(params, P.oracle_arch(cfg), record={"_compare": acts}), batch, "retinanet_losses", "RetinaNet-R50 2x800x1344")


def test_fcos_r50_full_size_matches_oracle():
    from basedet_amd.configs import FCOSConfig
    from basedet_amd.models import FCOS, params as P
    from basedet_amd.utils import DummyLoader
    from oracle.model import Oracle
    cfg = FCOSConfig()
    cfg.MODEL.BATCHSIZE = N
    params = P.init_fcos_params(cfg, seed=0, residual_gamma=0.25)
    rng = np.random.default_rng(7)
    for k in list(params):
        if k.startswith("head.") and k.rsplit(".", 2)[-2] in ("1", "4", "7", "10") and k.endswith(".weight"):
            params[k] = rng.uniform(0.7, 1.3, params[k].shape).astype(np.float32)       # GroupNorm gamma
        if k == "head.scales":
            params[k] = rng.uniform(0.8, 1.2, params[k].shape).astype(np.float32)
        if k == "head.bbox_pred.bias":
            params[k] = np.full_like(params[k], 0.5)                                     # keep relu(bbox_pred * scale) alive
    batch = next(DummyLoader(N, SIZE, seed=0))
    batch["data"] = (batch["data"] * 255).astype(np.float32)
    model = FCOS(cfg, params=params)
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    ref, aux = orc.fcos_losses(batch)
    ref_grads = orc.grads(ref["total_loss"])
    out = model(batch)
    pl = model._cur
    assert pl.labels.numel() == N * 22400
    assert np.array_equal(pl.labels.cpu().numpy(), aux["labels"])
    assert np.array_equal(pl.gt_offsets.cpu().numpy(), aux["gt_offsets"])
    st = pl.stats.cpu().numpy()
    assert st[0] == aux["num_fg"] and abs(st[1] - aux["sum_ctr"]) / aux["sum_ctr"] < 1e-5
    for k in ("cls_loss", "reg_loss", "ctr_loss", "total_loss"):
        got, want = float(out[k]), float(ref[k].detach())
        assert abs(got - want) / abs(want) < 2e-2, (k, got, want)
    model.backward()
    torch.cuda.synchronize()
    got = model.reference_grads()
    acts = model.debug_activations()
    orc2 = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=acts)
    l2, _ = orc2.fcos_losses(batch)
    _check_grads(names, got, ref_grads, orc2.grads(l2["total_loss"]), "FCOS-R50 2x800x1344", *FCOS_MIN_COS)
    _check_forward_layers(Oracle(params, P.oracle_arch(cfg), record={"_compare": acts}), batch, "fcos_losses", "FCOS-R50 2x800x1344", bound=3e-2)


def test_faster_rcnn_r50_full_size_matches_oracle():
    """C4 at its configured sizes: R50-FPN, P2-P6, TRAIN_PREV_NMS_TOPK 2000, TRAIN_POST_NMS_TOPK 1000, NUM_ROIS 512, 256 sampled
    anchors; the random sampling keys are supplied by the caller to both sides."""
    from basedet_amd.configs import FasterRCNNConfig
    from basedet_amd.models import FasterRCNN, params as P
    from basedet_amd.utils import DummyLoader
    from oracle.model import Oracle
    cfg = FasterRCNNConfig()
    cfg.MODEL.BATCHSIZE = N
    assert (cfg.MODEL.RPN.TRAIN_PREV_NMS_TOPK, cfg.MODEL.RPN.TRAIN_POST_NMS_TOPK, cfg.MODEL.RCNN.NUM_ROIS) == (2000, 1000, 512)
    params = P.init_faster_rcnn_params(cfg, 0, residual_gamma=0.25)
    for k in ("rpn.rpn_cls_score.weight", "rpn.rpn_bbox_offsets.weight", "rcnn.pred_cls.weight", "rcnn.pred_delta.weight",
              "rcnn.fc1.weight", "rcnn.fc2.weight", "rpn.rpn_conv.weight"):
        params[k] = (params[k] * 3).astype(np.float32)        # scores / deltas away from 0: non-trivial top-k, NMS, sampling
    batch = next(DummyLoader(N, SIZE, seed=0))
    batch["data"] = (batch["data"] * 255).astype(np.float32)
    model = FasterRCNN(cfg, params=params)
    names = P.trainable_names(params, cfg.MODEL.BACKBONE.FREEZE_AT)
    pl = model._plan(N, SIZE[0], SIZE[1])
    assert pl.A_total == 268569 and pl.rois.shape[1] == 1000
    Gmax = batch["gt_boxes"].shape[1]
    rng = np.random.default_rng(5)
    keys = dict(rpn_pos=rng.random((N, pl.A_total), dtype=np.float32), rpn_neg=rng.random((N, pl.A_total), dtype=np.float32),
                rcnn_fg=rng.random((N, pl.rois.shape[1] + Gmax), dtype=np.float32),
                rcnn_bg=rng.random((N, pl.rois.shape[1] + Gmax), dtype=np.float32))
    batch = dict(batch, sample_keys=keys)
    out = model(batch)
    model.backward()
    torch.cuda.synchronize()
    dbg = model.debug_samples()
    # (1) plain fp32 oracle: the RPN targets do not depend on the network output -> bit-exact; RPN losses within bf16 tolerance
    orc = Oracle(params, P.oracle_arch(cfg), trainable=names)
    ref, aux = orc.faster_rcnn_losses(batch, keys)
    ref_grads = orc.grads(ref["total_loss"])
    assert np.array_equal(dbg["rpn_labels"], aux["rpn_labels"])
    for k in ("rpn_cls_loss", "rpn_reg_loss"):
        got, want = float(out[k]), float(ref[k].detach())
        assert abs(got - want) / abs(want) < 2e-2, (k, got, want)
    # (2) oracle on the stored activations of the HIP run: identical scores -> identical proposals and samples
    acts = model.debug_activations()
    valid = dbg["s_labels"].reshape(-1) >= 0
    ch = cfg.MODEL.FPN.OUT_CHANNELS
    pooled = acts.pop("pooled")[valid]
    acts["pooled"] = pooled.reshape(-1, 49, ch).permute(0, 2, 1).reshape(-1, ch * 49).contiguous()
    for k in ("fc1", "fc2", "rcnn_raw"):
        acts[k] = acts[k][valid].contiguous()
    orc2 = Oracle(params, P.oracle_arch(cfg), trainable=names, sim_bf16=True, inject=acts)
    l2, aux2 = orc2.faster_rcnn_losses(batch, keys)
    for n in range(N):
        m = int(dbg["num_rois"][n])
        assert m == len(aux2["rois"][n])
        np.testing.assert_allclose(dbg["rois"][n, :m], aux2["rois"][n], rtol=1e-5, atol=1e-3)
    assert int(valid.sum()) == len(aux2["s_labels"]) == N * 512
    assert np.array_equal(dbg["s_labels"].reshape(-1)[valid], aux2["s_labels"])
    np.testing.assert_allclose(dbg["s_rois"].reshape(-1, 4)[valid], aux2["s_rois"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(dbg["s_targets"].reshape(-1, 4)[valid], aux2["s_targets"], rtol=1e-3, atol=1e-3)
    assert (aux2["s_labels"] > 0).sum() >= N
    for k in ("rpn_cls_loss", "rpn_reg_loss", "rcnn_cls_loss", "rcnn_reg_loss", "total_loss"):
        got, want = float(out[k]), float(l2[k].detach())
        assert abs(got - want) / abs(want) < 2e-2, (k, got, want)
    # (3) every parameter gradient: direction against the plain oracle (as RetinaNet / FCOS above), rel-L2 against the injected one (the
    # box head's gradients pass RoIAlign's tiled backward: fp32 sums per 8 x 8 tile, ONE rounding to bf16 when the tile is written on top
    # of the RPN head's bf16 dL/dP; observed worst 2.3e-3 -- the bound of rounds 3-5, 3e-2, dated from the packed-bf16 atomics)
    _check_grads(names, model.reference_grads(), ref_grads, orc2.grads(l2["total_loss"]), "Faster R-CNN R50 2x800x1344, 2000/1000/512",
                 *FRCNN_MIN_COS, inj_bound=1e-2)
