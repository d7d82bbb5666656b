"""Built-in inference recipes: what the episodic MetaFCOS models of the reference's two target benchmarks set ON TOP of
``get_default_cfg()``, restricted to keys the inference path reads (sylph_amd.engine.config_from_cfg, modeling, runner).

``sylph://<name>`` resolves to a yaml under ``$SYLPH_CONFIG_ROOT`` first -- point it at the reference's own ``configs/``
directory and its files load unchanged (training keys included, tests/test_host_cpu.py) -- and only falls back to the
recipe of that name here.  The recipes are deltas in Python, not files: nothing about training (solver, datasets,
mappers, loss switches, freeze flags) is restated.

Sources for the values: sylph/runner/meta_fcos_runner.py:47-171 (defaults), the reference's
configs/{COCO-Detection,LVISv1-Detection}/Meta-FCOS/*.yaml recipes (MODEL.FCOS / MODEL.META_LEARN blocks).
"""
import copy

_EPISODIC_CODE_GENERATOR = {
    "NAME": "CodeGenerator",
    "USE_MASK": True, "ALL_MASK": False, "MASK_NORM": "GN",
    "TOWER_LAYERS": [["GN", "ReLU"], ["GN", "ReLU"]],
    "CLS_LAYER": ["", "", 1], "BIAS_LAYER": ["", "", 1],
    "USE_BIAS": True, "CONV_L2_NORM": True,
    "OUT_CHANNEL": 256, "POST_NORM": "GN",
}

_R50_FPN_P3_P7 = {
    "BACKBONE": {"NAME": "build_fcos_resnet_fpn_backbone"},
    "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res3", "res4", "res5"]},
    "FPN": {"IN_FEATURES": ["res3", "res4", "res5"]},
}


def _episodic_model(num_classes, fcos=None, code_generator=None):
    cg = dict(_EPISODIC_CODE_GENERATOR, **(code_generator or {}))
    model = copy.deepcopy(_R50_FPN_P3_P7)
    model.update({
        "META_ARCHITECTURE": "MetaOneStageDetector",
        "PROPOSAL_GENERATOR": {"NAME": "MetaFCOS"},
        "FCOS": dict({"NUM_CLASSES": num_classes, "BOX_QUALITY": ["ctrness"]}, **(fcos or {})),
        "META_LEARN": {"EPISODIC_LEARNING": True, "CLASS": 3, "SHOT": 5, "EVAL_SHOT": 10, "BASE_EVAL_SHOT": 10, "QUERY_SHOT": 1,
                       "CODE_GENERATOR": cg},
    })
    return {"MODEL": model, "TEST": {"REPEAT_TEST": 5}}


RECIPES = {
    # COCO few-shot: 60 base classes at meta-training time, 20 novel; GN/L2/scale class codes
    "COCO-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml": _episodic_model(60),
    # LVIS v1: 866 frequent + common base classes, 300 detections per image, the bias code is L2-normalised too
    "LVISv1-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml": _episodic_model(
        866,
        fcos={"POST_NMS_TOPK_TEST": 300, "NUM_CLS_CONVS": 4, "CLS_LOGITS_KERNEL_SIZE": 1, "NORM": "GN"},
        code_generator={"BIAS_L2_NORM": True, "USE_WEIGHT_SCALE": True, "USE_PER_CLS_SCALE": True,
                        "ROI_BOX": {"POOLER_RESOLUTION": 7, "POOLER_TYPE": "ROIAlignV2"}}),
    # LVIS v1 with the ROIEncoder hyper-network (configs/LVISv1-Detection/Meta-FCOS/Meta-FCOS-ROI-Encoder-finetune.yaml on top of
    # Base-Meta-FCOS.yaml): transformer code generator, CondConvBlock head; run it with MetaFCOSROIEncoderRunner
    "LVISv1-Detection/Meta-FCOS/Meta-FCOS-ROI-Encoder-finetune.yaml": {
        "MODEL": dict(copy.deepcopy(_R50_FPN_P3_P7), **{
            "META_ARCHITECTURE": "MetaOneStageDetector",
            "PROPOSAL_GENERATOR": {"NAME": "MetaFCOS", "OWD": False},
            "FCOS": {"NUM_CLASSES": 1103, "POST_NMS_TOPK_TEST": 300, "NUM_CLS_CONVS": 4, "CLS_LOGITS_KERNEL_SIZE": 1, "NORM": "GN",
                     "BOX_QUALITY": ["ctrness"]},
            "META_LEARN": {"EPISODIC_LEARNING": True, "CLASS": 3, "SHOT": 5, "EVAL_SHOT": 10, "QUERY_SHOT": 1,
                           "CODE_GENERATOR": {
                               "NAME": "ROIEncoder",
                               "ROI_BOX": {"POOLER_RESOLUTION": 7, "POOLER_TYPE": "ROIAlignV2"},
                               "TOKENIZER": {"NUM_CONV": 2, "CONV_DIM": 256, "NORM": "GN", "NUM_FC": 2, "FC_DIM": 256},
                               "TRANSFORMER_ENCODER": {"LAYERS": 2, "HEADS": 8, "DROPOUT": 0.1},
                               "HEAD": {"NUM_FC": 2, "FC_DIM": 512, "OUTPUT_DIM": 256}}},
        }),
    },
}


def get_recipe(name: str):
    """Deep copy of the built-in recipe called `name` (the part after ``sylph://``), or None."""
    r = RECIPES.get(name)
    return copy.deepcopy(r) if r is not None else None
