"""Config surface of the Sylph inference path: a small yacs-style node, the defaults the path
reads, ``_BASE_`` inheritance and the ``sylph://`` prefix.

Mirrors (reference paths relative to /root/reference):
  * sylph/config/config.py:20-65        CfgNode.merge_from_file, ``sylph://`` -> $SYLPH_CONFIG_ROOT, else sylph_amd.recipes,
                                        ``_BASE_`` rerouting
  * sylph/runner/adet_configs.py:25-61  MODEL.FCOS.* defaults
  * sylph/runner/default_configs.py:8-167  DATASETS.*, MODEL.BACKBONE.FREEZE*, MODEL.PROPOSAL_GENERATOR.*,
                                        MODEL.FCOS.{BOX_QUALITY,...}, MODEL.TFA.*, MODEL.META_LEARN.*,
                                        CODE_GENERATOR.* and the ROIEncoder sub-nodes, TEST.REPEAT_TEST
  * sylph/runner/meta_fcos_runner.py:104-114  get_default_cfg composition
The detectron2/d2go base node is not available here, so only the base keys this path reads are
defaulted; any other key found in a yaml (SOLVER.*, D2GO_DATA.*, DATALOADER.*, ...) is accepted
and stored as-is ("unknown keys are tolerated").
"""
import copy
import os
from typing import Any, List

import yaml

BASE_KEY = "_BASE_"
SYLPH_PREFIX = "sylph://"


def config_roots() -> List[str]:
    """Directories searched for ``sylph://<rel>``: $SYLPH_CONFIG_ROOT (':'-separated; e.g. the reference's own
    ``configs/`` directory, whose yamls load unchanged).  Names found in no root fall back to the built-in inference
    recipes of sylph_amd.recipes."""
    return [r for r in os.environ.get("SYLPH_CONFIG_ROOT", "").split(":") if r]


def reroute_config_path(path: str) -> str:
    """sylph/config/config.py:32-42.  A ``sylph://`` name that exists under no config root is returned unchanged
    (load_yaml_with_base then looks it up among the built-in recipes)."""
    if path.startswith(SYLPH_PREFIX):
        rel = path[len(SYLPH_PREFIX):]
        for root in config_roots():
            cand = os.path.join(root, rel)
            if os.path.exists(cand):
                return cand
    return path


class CfgNode(dict):
    """Attribute-access dict with yacs-like merge semantics."""

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {name} on a frozen CfgNode")
        self[name] = value

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def clone(self):
        c = copy.deepcopy(self)
        return c

    def __deepcopy__(self, memo):
        c = CfgNode()
        for k, v in self.items():
            dict.__setitem__(c, k, copy.deepcopy(v, memo))
        object.__setattr__(c, "_frozen", self.is_frozen())
        return c

    # ---- loading -------------------------------------------------------------------------
    @staticmethod
    def load_yaml_with_base(filename: str) -> dict:
        filename = reroute_config_path(filename)
        if filename.startswith(SYLPH_PREFIX):
            from .recipes import get_recipe
            recipe = get_recipe(filename[len(SYLPH_PREFIX):])
            if recipe is None:
                raise FileNotFoundError(f"{filename}: not under $SYLPH_CONFIG_ROOT and not a built-in recipe")
            return recipe
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            base = reroute_config_path(base)
            if not base.startswith("/") and not base.startswith("~") and not base.startswith(SYLPH_PREFIX):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dict(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, cfg_filename: str):
        loaded = CfgNode.load_yaml_with_base(cfg_filename)
        self.merge_from_other_cfg(loaded)

    def merge_from_other_cfg(self, other: dict):
        _merge_node(other, self)

    def merge_from_list(self, opts: List[Any]):
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            if isinstance(val, str):
                try:
                    val = yaml.safe_load(val)
                except yaml.YAMLError:
                    pass
            node[parts[-1]] = CfgNode(val) if isinstance(val, dict) else val

    def dump(self) -> str:
        return yaml.safe_dump(_to_plain(self))


def _to_plain(n):
    if isinstance(n, dict):
        return {k: _to_plain(v) for k, v in n.items()}
    if isinstance(n, tuple):
        return list(n)
    return n


def _merge_dict(src: dict, dst: dict):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dict(v, dst[k])
        else:
            dst[k] = v


def _merge_node(src: dict, dst: CfgNode):
    for k, v in src.items():
        if isinstance(v, dict):
            if k not in dst or not isinstance(dst[k], CfgNode):
                dst[k] = CfgNode()
            _merge_node(v, dst[k])
        else:
            if isinstance(v, str) and v.startswith("(") and v.endswith(")"):
                try:  # yacs accepts python tuples written as strings, e.g. STEPS: (60000, 80000)
                    v = tuple(yaml.safe_load("[" + v[1:-1] + "]"))
                except yaml.YAMLError:
                    pass
            dst[k] = v


_NEG_LOG_99 = None

DEFAULTS = {
    "VERSION": 2,
    "SEED": -1,
    "OUTPUT_DIR": "./output",
    "MODEL": {
        "DEVICE": "cuda",
        "WEIGHTS": "",
        "WEIGHTS_FILTER_BY_MODULE": [],
        "META_ARCHITECTURE": "MetaOneStageDetector",
        "PIXEL_MEAN": [103.530, 116.280, 123.675],
        "PIXEL_STD": [1.0, 1.0, 1.0],
        "MOBILENET": False,
        "DDP_FIND_UNUSED_PARAMETERS": False,
        "BACKBONE": {"NAME": "build_fcos_resnet_fpn_backbone", "FREEZE_AT": 2, "ANTI_ALIAS": False,
                     "FREEZE": False, "FREEZE_EXCLUDE": []},
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res3", "res4", "res5"], "NUM_GROUPS": 1,
                    "NORM": "FrozenBN", "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True, "RES5_DILATION": 1,
                    "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64, "DEFORM_INTERVAL": 1,
                    "DEFORM_ON_PER_STAGE": [False, False, False, False]},
        "FPN": {"IN_FEATURES": ["res3", "res4", "res5"], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"},
        "PROPOSAL_GENERATOR": {"NAME": "MetaFCOS", "MIN_SIZE": 0, "OWD": False, "FREEZE_CLS_TOWER": False,
                               "FREEZE_CLS_LOGITS": False, "FREEZE_BBOX_BRANCH": False,
                               "FREEZE_BBOX_TOWER": False, "FREEZE": False},
        "ROI_HEADS": {"FREEZE": False},
        "FCOS": {
            "NUM_CLASSES": 80, "IN_FEATURES": ["p3", "p4", "p5", "p6", "p7"],
            "FPN_STRIDES": [8, 16, 32, 64, 128], "PRIOR_PROB": 0.01,
            "INFERENCE_TH_TRAIN": 0.05, "INFERENCE_TH_TEST": 0.05, "NMS_TH": 0.6,
            "PRE_NMS_TOPK_TRAIN": 1000, "PRE_NMS_TOPK_TEST": 1000,
            "POST_NMS_TOPK_TRAIN": 100, "POST_NMS_TOPK_TEST": 100, "TOP_LEVELS": 2, "NORM": "GN",
            "USE_SCALE": True, "THRESH_WITH_CTR": False, "LOSS_ALPHA": 0.25, "LOSS_GAMMA": 2.0,
            "SIZES_OF_INTEREST": [64, 128, 256, 512], "USE_RELU": True, "USE_DEFORMABLE": False,
            "NUM_CLS_CONVS": 4, "NUM_BOX_CONVS": 4, "NUM_SHARE_CONVS": 0, "CENTER_SAMPLE": True,
            "POS_RADIUS": 1.5, "LOC_LOSS_TYPE": "giou", "YIELD_PROPOSAL": False,
            "BOX_QUALITY": ["ctrness"], "IOU_MASK": False, "CLS_LOGITS_KERNEL_SIZE": 1,
            "L2_NORM_CLS_WEIGHT": False,
        },
        "TFA": {"FINETINE": False, "TRAIN_SHOT": 10, "USE_PRETRAINED_BASE_CLS_LOGITS": True,
                "EVAL_WITH_PRETRAINED_BASE_CLS_LOGITS": False},
        "META_LEARN": {
            "EPISODIC_LEARNING": False, "SHOT": 5, "EVAL_SHOT": 10, "BASE_EVAL_SHOT": 10, "CLASS": 5,
            "USE_ALL_GTS_IN_BASE_CLASSES": True, "EVAL_WITH_PRETRAINED_CODE": False, "QUERY_SHOT": 1,
            "CODE_GENERATOR": {
                "FREEZE": False, "DISTILLATION_LOSS_WEIGHT": 0.0, "NAME": "CodeGenerator",
                "ROI_BOX": {"POOLER_RESOLUTION": 7, "POOLER_TYPE": "ROIAlignV2",
                            "FPN_MULTILEVEL_FEATURE": False},
                "USE_MASK": True, "ALL_MASK": False, "MASK_NORM": "GN", "CONV_L2_NORM": False,
                "USE_BIAS": True, "BIAS_L2_NORM": False, "TOWER_LAYERS": [["GN", ""]],
                "CLS_LAYER": ["GN", "", 1], "USE_WEIGHT_SCALE": True, "BIAS_LAYER": [],
                "WEIGHT_LAYER": [], "SCALE_LAYER": [], "BOX_ON": False, "BOX_TOWER_LAYERS": [],
                "BOX_CLS_LAYER": ["", "", 2], "BOX_BIAS_LAYER": [], "CONTRASTIVE_LOSS": "",
                "INIT_NORM_LAYER": False, "CLS_REWEIGHT": False, "META_WEIGHT": False,
                "META_BIAS": False, "USE_PER_CLS_SCALE": False, "COMPRESS_CODE_W_MAX": False,
                "POST_NORM": "GN", "IN_CHANNEL": 256, "OUT_CHANNEL": 256, "USE_DEFORMABLE": False,
                "TOKENIZER": {"NUM_CONV": 0, "CONV_DIM": 256, "NORM": "", "NUM_FC": 1, "FC_DIM": 256},
                "TRANSFORMER_ENCODER": {"LAYERS": 1, "HEADS": 8, "DROPOUT": 0.1},
                "HEAD": {"NUM_FC": 1, "FC_DIM": 512, "OUTPUT_DIM": 256},
            },
        },
    },
    "INPUT": {"MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR", "HFLIP_TRAIN": True,
              "MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "CROP": {"CROP_INSTANCE": True}},
    "DATASETS": {"TRAIN": (), "TEST": (), "ID_TRAIN": [0], "ID_TEST": [0], "BASE_CLASSES_SPLIT": "",
                 "NOVEL_CLASSES_SPLIT": "", "NUMS_CLASSES": [0]},
    "DATALOADER": {"NUM_WORKERS": 4, "ASPECT_RATIO_GROUPING": True},
    "SOLVER": {"IMS_PER_BATCH": 16, "MAX_ITER": 40000},
    "TEST": {"EVAL_PERIOD": 0, "REPEAT_TEST": 1, "DETECTIONS_PER_IMAGE": 100},
}


def get_default_cfg() -> CfgNode:
    """The node MetaFCOSRunner.get_default_cfg() returns (meta_fcos_runner.py:104-114)."""
    return CfgNode(copy.deepcopy(DEFAULTS))


def get_roi_encoder_default_cfg() -> CfgNode:
    """sylph/runner/meta_fcos_roi_encoder_runner.py: ROIEncoder code generator defaults."""
    cfg = get_default_cfg()
    cfg.MODEL.META_LEARN.CODE_GENERATOR.NAME = "ROIEncoder"
    return cfg
