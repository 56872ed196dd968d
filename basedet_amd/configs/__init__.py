"""Minimal config tree that keeps the reference's names and default values for the hot path.

Mirrors basedet/configs/extra_cfg.py (ModelConfig/SolverConfig/TrainerConfig/TestConfig defaults),
basedet/configs/det_model/retinanet_cfg.py:5-49 and fcos_cfg.py:7-56.  The reference's ConfigDict comes from
the un-vendored `basecore`; this one supports attribute access, nested `merge`, `get`, and key-value `opts`.
"""
import copy


class ConfigDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            v = ConfigDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other):
        """basecore ConfigDict.merge: recursive update; also accepts a flat [KEY, VALUE, ...] opts list
        (basedet/tools/det_train.py:58-63,71)."""
        if isinstance(other, (list, tuple)):
            assert len(other) % 2 == 0
            for k, v in zip(other[0::2], other[1::2]):
                node = self
                parts = k.split(".")
                for p in parts[:-1]:
                    node = node[p]
                old = node.get(parts[-1])
                if isinstance(v, str) and old is not None and not isinstance(old, str):
                    import ast
                    v = ast.literal_eval(v)
                node[parts[-1]] = v
            return self
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = copy.deepcopy(v)
        return self


class DetectionConfig(ConfigDict):
    """configs/detection_cfg.py:23-106: the sections every model config shares, plus the builder methods the training entry
    (tools/det_train.py:111) calls.  `DetectionConfig(cfg=None, **kwargs)`: later values win, as in the reference."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.MODEL = dict(
            NAME="", BATCHSIZE=2, WEIGHTS=None,
            BACKBONE=dict(NAME="resnet50", IMG_MEAN=[103.530, 116.280, 123.675], IMG_STD=[57.375, 57.12, 58.395],
                          NORM="FrozenBN", FREEZE_AT=2),                                   # extra_cfg.py:46-57
        )
        self.DATA = dict(BUILDER_NAME="DataloaderBuilder", NUM_CLASSES=80, NUM_WORKERS=2, ENABLE_INFINITE_SAMPLER=True,
                         TRAIN=dict(name="coco_2017_train", remove_images_without_annotations=True,
                                    order=("image", "boxes", "boxes_category", "info")),
                         TEST=dict(name="coco_2017_val", remove_images_without_annotations=False, order=("image", "info")))
        self.SOLVER = dict(BUILDER_NAME="DetSolver", OPTIMIZER_NAME="SGD", LR_SCHEDULER_NAME="MultiStepLR",
                           BASIC_LR=0.01 / 16.0, WEIGHT_DECAY=1e-4, EXTRA_OPT_ARGS=dict(momentum=0.9),
                           REDUCE_MODE="MEAN", EPOCHWISE_STEP=False, WARM_ITERS=500, NUM_IMAGE_PER_EPOCH=80000,
                           MAX_EPOCH=18, LR_DECAY_STAGES=[12, 16], LR_DECAY_RATE=0.1,
                           EXTRA_LR_ARGS=dict())                                           # extra_cfg.py:60-78
        self.TRAINER = dict(NAME="DetTrainer", RESUME=False, AMP=dict(ENABLE=False, DYNAMIC_SCALE=False),
                            EMA=dict(ENABLE=False, ALPHA=5e-4, MOMENTUM=None, UPDATE_PERIOD=1, BURNIN_ITER=2000),
                            GRAD_CLIP=dict(ENABLE=False, TYPE="value", ARGS=dict(lower=-1, upper=1)))
        self.HOOKS = dict(BUILDER_NAME="SimpleHookList")
        self.TEST = dict(EVALUATOR_NAME="COCOEvaluator", IOU_THRESHOLD=0.5, CLS_THRESHOLD=0.05, MAX_BOXES_PER_IMAGE=100,
                         IMG_MIN_SIZE=800, IMG_MAX_SIZE=1333, VIS_THRESHOLD=0.3, EVAL_EPOCH_INTERVAL=None)
        self.AUG = dict(TRAIN_VALUE=(
            ("MGE_ShortestEdgeResize", dict(min_size=(640, 672, 704, 736, 768, 800), max_size=1333, sample_style="choice")),
            ("MGE_RandomHorizontalFlip", dict(prob=0.5)),
            ("MGE_ToMode", dict(mode="CHW"))),
            TRAIN_WRAPPER=(("MGE_Compose", dict(order=("image", "boxes", "boxes_category"))),))
        self.GLOBAL = dict(OUTPUT_DIR="logs", CKPT_SAVE_DIR="/data/Outputs/model_logs/basedet_playground", LOG_INTERVAL=20,
                           TENSORBOARD=dict(ENABLE=False))                                 # extra_cfg.py:35-44
        if cfg:
            self.merge(cfg)
        if kwargs:
            self.merge(kwargs)

    def _override(self, cfg, kwargs):
        if cfg:
            self.merge(cfg)
        if kwargs:
            self.merge(kwargs)

    # builders (detection_cfg.py:55-106) ------------------------------------------------------------------
    def build_model(self):
        from ..utils.registry import registers
        from .. import models  # noqa: F401  (fills the registry)
        return registers.models.get(self.MODEL.NAME)(self)

    def build_solver(self, model):
        from ..utils.registry import registers
        from .. import solver  # noqa: F401
        return registers.solvers.get(self.SOLVER.BUILDER_NAME).build(self, model)

    def build_dataloader(self):
        """The reference builds a COCO reader here (registers.dataloader, data/build.py) -- out of the hot-path scope.  The
        synthetic loader of the reference's own benchmark harness (utils/dummy.py, tools/benchmark.py:173) stands in."""
        from ..utils import DummyLoader
        from .. import comm
        if self.DATA.get("BUILDER_NAME") != "DummyLoader":
            # a playground config names a COCO reader (DATA.TRAIN.name, AUG, NUM_WORKERS): silently training on noise instead would
            # look like a successful run.  The caller has to opt into synthetic data explicitly.
            raise RuntimeError(
                f"DATA.BUILDER_NAME = {self.DATA.get('BUILDER_NAME')!r}: this build has no dataset readers (hot-path scope, DESIGN.md); "
                "the only data source is the synthetic DummyLoader of the reference's benchmark harness.  Set DATA.BUILDER_NAME = "
                "'DummyLoader' (basedet_train: --synthetic) to train on it, or hand your own iterable of batch dicts to DetTrainer.")
        return DummyLoader(self.MODEL.BATCHSIZE, tuple(self.DATA.get("DUMMY_SIZE", (800, 1344))), seed=comm.rank())

    def build_trainer(self):
        """detection_cfg.py:67-106: model (+ weights), parameter broadcast, dataloader, solver -> trainer."""
        from ..engine import DetTrainer
        from ..solver import broadcast_parameters
        model = self.build_model()
        if self.MODEL.WEIGHTS:
            model.load_weights(self.MODEL.WEIGHTS)
        broadcast_parameters(model)
        return DetTrainer(self, model, self.build_dataloader(), self.build_solver(model))


def _base():
    return DetectionConfig()


class RetinaNetConfig(DetectionConfig):
    """basedet/configs/det_model/retinanet_cfg.py:5-56."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.merge(dict(MODEL=dict(
            NAME="RetinaNet",
            BACKBONE=dict(OUT_FEATURES=["res3", "res4", "res5"], OUT_FEATURE_CHANNELS=[512, 1024, 2048]),
            FPN=dict(OUT_FEATURES=["p3", "p4", "p5", "p6", "p7"], NORM=None, STRIDES=[8, 16, 32, 64, 128],
                     TOP_BLOCK_IN_CHANNELS=2048, TOP_BLOCK_IN_FEATURE="res5", OUT_CHANNELS=256),
            ANCHOR=dict(SCALES=[[x, x * 2 ** (1.0 / 3), x * 2 ** (2.0 / 3)] for x in [32, 64, 128, 256, 512]],
                        RATIOS=[[0.5, 1, 2]], OFFSET=0.5),
            LOSSES=dict(FOCAL_LOSS_ALPHA=0.25, FOCAL_LOSS_GAMMA=2, SMOOTH_L1_BETA=0.0, REG_LOSS_WEIGHT=1.0),
            BOX_REG=dict(MEAN=[0.0, 0.0, 0.0, 0.0], STD=[1.0, 1.0, 1.0, 1.0]),
            MATCHER=dict(THRESHOLDS=[0.4, 0.5], LABELS=[0, -1, 1], ALLOW_LOW_QUALITY=True),
            HEAD=dict(NUM_CONVS=4, CLS_PRIOR_PROB=0.01),
        )))
        self._override(cfg, kwargs)


class FreeAnchorConfig(RetinaNetConfig):
    """basedet/configs/det_model/freeanchor_cfg.py:5-33.  The reference writes the key FOCLA_LOSS_ALPHA (:10), so the value the
    model reads, FOCAL_LOSS_ALPHA (free_anchor.py:132), keeps RetinaNet's 0.25; the stray key is carried as written."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.merge(dict(MODEL=dict(
            NAME="FreeAnchor",
            LOSSES=dict(FOCLA_LOSS_ALPHA=0.5, FOCAL_LOSS_GAMMA=2, SMOOTH_L1_BETA=0.0, REG_LOSS_WEIGHT=0.75),
            BOX_REG=dict(STD=[0.1, 0.1, 0.2, 0.2]),
            HEAD=dict(CLS_PRIOR_PROB=0.02),
            BUCKET=dict(BOX_IOU_THRESH=0.6, BUCKET_SIZE=50),
        )))
        self._override(cfg, kwargs)


def retinanet_r18_config():
    """BASELINE.json configs[0]: RetinaNet-R18-FPN."""
    cfg = RetinaNetConfig()
    cfg.merge(dict(MODEL=dict(BACKBONE=dict(NAME="resnet18", OUT_FEATURE_CHANNELS=[128, 256, 512]),
                              FPN=dict(TOP_BLOCK_IN_CHANNELS=512))))
    return cfg


class FCOSConfig(DetectionConfig):
    """basedet/configs/det_model/fcos_cfg.py:7-56."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.merge(dict(MODEL=dict(
            NAME="FCOS",
            ANCHOR=dict(NUM_ANCHORS=1, OFFSET=0.5),
            BACKBONE=dict(OUT_FEATURES=["res3", "res4", "res5"], OUT_FEATURE_CHANNELS=[512, 1024, 2048]),
            FPN=dict(OUT_FEATURES=["p3", "p4", "p5", "p6", "p7"], NORM=None, STRIDES=[8, 16, 32, 64, 128],
                     TOP_BLOCK_IN_CHANNELS=2048, OUT_CHANNELS=256, TOP_BLOCK_IN_FEATURE="res5"),
            LOSSES=dict(FOCAL_LOSS_ALPHA=0.25, FOCAL_LOSS_GAMMA=2, IOU_LOSS_TYPE="giou", REG_LOSS_WEIGHT=1.0),
            BOX_REG=dict(MEAN=[0.0, 0.0, 0.0, 0.0], STD=[1.0, 1.0, 1.0, 1.0]),
            HEAD=dict(NUM_CONVS=4, CLS_PRIOR_PROB=0.01,
                      OBJECT_SIZES_OF_INTEREST=[[-1, 64], [64, 128], [128, 256], [256, 512], [512, float("inf")]],
                      CENTER_SAMPLING_RADIUS=1.5),
        ), TEST=dict(IOU_THRESHOLD=0.6)))
        self._override(cfg, kwargs)


class FasterRCNNConfig(DetectionConfig):
    """basedet/configs/det_model/faster_rcnn_cfg.py:5-78."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.merge(dict(MODEL=dict(
            NAME="FasterRCNN",
            BACKBONE=dict(OUT_FEATURES=["res2", "res3", "res4", "res5"], OUT_FEATURE_CHANNELS=[256, 512, 1024, 2048]),
            FPN=dict(OUT_FEATURES=["p2", "p3", "p4", "p5", "p6"], NORM=None, STRIDES=[4, 8, 16, 32, 64],
                     TOP_BLOCK_IN_CHANNELS=2048, OUT_CHANNELS=256, TOP_BLOCK_IN_FEATURE="p5"),
            RPN=dict(CHANNELS=256, NMS_THRESHOLD=0.7, NUM_SAMPLE_ANCHORS=256, POSITIVE_ANCHOR_RATIO=0.5,
                     TRAIN_PREV_NMS_TOPK=2000, TRAIN_POST_NMS_TOPK=1000, TEST_PREV_NMS_TOPK=1000, TEST_POST_NMS_TOPK=1000),
            ROI_POOLER=dict(METHOD="roi_align", SIZE=(7, 7)),
            RCNN=dict(IN_FEATURES=["p2", "p3", "p4", "p5"], STRIDES=[4, 8, 16, 32], NUM_ROIS=512, FG_RATIO=0.5,
                      FG_THRESHOLD=0.5, BG_THRESHOLD_HIGH=0.5, BG_THRESHOLD_LOW=0.0),
            ANCHOR=dict(SCALES=[[x] for x in [32, 64, 128, 256, 512]], RATIOS=[[0.5, 1, 2]], OFFSET=0.5),
            LOSSES=dict(RPN_SMOOTH_L1_BETA=0, RCNN_SMOOTH_L1_BETA=0),
            RPN_BOX_REG=dict(MEAN=[0.0, 0.0, 0.0, 0.0], STD=[1.0, 1.0, 1.0, 1.0]),
            RCNN_BOX_REG=dict(MEAN=[0.0, 0.0, 0.0, 0.0], STD=[0.1, 0.1, 0.2, 0.2]),
            MATCHER=dict(THRESHOLDS=[0.3, 0.7], LABELS=[0, -1, 1], ALLOW_LOW_QUALITY=True),
        ), SOLVER=dict(BASIC_LR=0.02 / 16, WARM_ITERS=500, MAX_EPOCH=18, LR_DECAY_STAGES=[12, 16])))
        self._override(cfg, kwargs)


class ATSSConfig(FCOSConfig):
    """basedet/configs/det_model/atss_cfg.py:5-26 (FCOS with the ATSS assignment; the size-of-interest / centre-sampling keys go)."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.merge(dict(MODEL=dict(NAME="ATSS", ANCHOR=dict(SCALE=8, TOPK=9), LOSSES=dict(REG_LOSS_WEIGHT=2.0))))
        del self.MODEL.HEAD["OBJECT_SIZES_OF_INTEREST"]
        del self.MODEL.HEAD["CENTER_SAMPLING_RADIUS"]
        self._override(cfg, kwargs)


class OTAConfig(FCOSConfig):
    """basedet/configs/det_model/ota_cfg.py:6-14: FCOS with the OTAPointHead flags and the top-k ("simOTA") matcher."""

    def __init__(self, cfg=None, **kwargs):
        super().__init__()
        self.merge(dict(MODEL=dict(NAME="OTA", MATCHING="topk", HEAD=dict(WITH_NORM=True, SHARE_PARAM=True, NORM_REG_TARGETS=True))))
        self._override(cfg, kwargs)
