"""Host-side mirror of the reference's meta-architecture for the inference path.

    model(batched_inputs, class_code=None, run_type=None)

keeps the reference's dispatch, argument meaning and error behaviour
(sylph/modeling/meta_arch/meta_one_stage_detector.py:415-455) while every tensor op runs in
libsylph_hip (one C-ABI call per stage, see sylph_amd/engine.py).  The registries resolve the yaml
names (MODEL.META_ARCHITECTURE "MetaOneStageDetector", MODEL.BACKBONE.NAME
"build_fcos_resnet_fpn_backbone", MODEL.PROPOSAL_GENERATOR.NAME "MetaFCOS",
MODEL.META_LEARN.CODE_GENERATOR.NAME "CodeGenerator" / "ROIEncoder"; sylph/modeling/code_generator/build.py:18-39) to
components the meta-architecture builds, binds to its HIP context and calls where the reference calls the nn.Modules:
another implementation registered under another name plugs in the same way.
"""
import logging
from typing import Any, Dict, List, Optional

import torch
from torch import nn

from .engine import Engine
from .structures import Boxes, Instances

logger = logging.getLogger(__name__)


class Registry:
    """name -> object, with the detectron2 Registry surface (register() decorator, get())."""

    def __init__(self, name: str):
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    def register(self, obj: Any = None):
        if obj is None:
            def deco(o):
                self._obj_map[o.__name__] = o
                return o
            return deco
        self._obj_map[obj.__name__] = obj
        return obj

    def get(self, name: str) -> Any:
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
CODE_GENERATOR_REGISTRY = Registry("CODE_GENERATOR")


class HipComponent:
    """A registry entry of this package is a real (if thin) component: the meta-architecture builds it from the yaml name, binds it
    to its HIP context and CALLS it where the reference calls the corresponding nn.Module -- so another implementation registered
    under another name (MODEL.BACKBONE.NAME / PROPOSAL_GENERATOR.NAME / CODE_GENERATOR.NAME) is picked up the same way.  The
    tensors stay inside the context (libsylph_hip works on the "current batch"): a component receives host-side arguments and
    returns host-visible results."""

    engine: Optional[Engine] = None

    def bind(self, engine: Engine):
        self.engine = engine
        return self


class FCOSResNetFPNBackbone(HipComponent):
    """convert_batched_inputs_to_image_list + self.backbone(images.tensor) (meta_one_stage_detector.py:174-181,273): normalise,
    pad to size_divisibility, ResNet-FPN + P6/P7.  The pyramid stays in the context as the current batch."""

    size_divisibility = 32

    def __init__(self, cfg, input_shape=None):
        if list(cfg.MODEL.FPN.IN_FEATURES) != ["res3", "res4", "res5"]:
            raise NotImplementedError("FPN.IN_FEATURES must be [res3, res4, res5]")
        if int(cfg.MODEL.FCOS.TOP_LEVELS) != 2:
            raise NotImplementedError("MODEL.FCOS.TOP_LEVELS must be 2 (P6, P7 from p5)")
        if str(cfg.MODEL.FPN.get("NORM", "")) != "":
            raise NotImplementedError("MODEL.FPN.NORM is not supported")
        self.depth = int(cfg.MODEL.RESNETS.DEPTH)

    def __call__(self, images: Optional[List[torch.Tensor]] = None, images_u8: Optional[List[torch.Tensor]] = None,
                 resize_hw=None, rgb_input: bool = False):
        """(3,H,W) float BGR images, or uint8 HWC originals + their ResizeShortestEdge targets (fused input pipeline).
        Returns the per-image (h, w) the network saw (ImageList.image_sizes)."""
        eng = self.engine
        if images_u8 is not None:
            sizes = [(int(s[0]), int(s[1])) for s in resize_hw]
            eng.preprocess_u8(images_u8, sizes, rgb_input=rgb_input)
        else:
            eng.preprocess(images)
            sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        eng.backbone()
        return sizes


@BACKBONE_REGISTRY.register()
def build_fcos_resnet_fpn_backbone(cfg, input_shape=None):
    """adet backbone/fpn.py build_fcos_resnet_fpn_backbone as named by the yamls."""
    return FCOSResNetFPNBackbone(cfg, input_shape)


@PROPOSAL_GENERATOR_REGISTRY.register()
class MetaFCOS(HipComponent):
    """MetaFCOS.forward at inference (meta_fcos/fcos.py:140-260): towers + class-conditional classifier on the current pyramid,
    predict_proposals, NMS, detector_postprocess to the requested output sizes."""

    def __init__(self, cfg, input_shape=None):
        pass

    def __call__(self, cls_conv: torch.Tensor, cls_bias: Optional[torch.Tensor], out_sizes, raw: bool = False):
        self.engine.head(cls_conv, cls_bias, raw=raw)
        return self.engine.decode(out_sizes)

    def forward_pretrained(self, out_sizes):
        """support_set_per_class_code = None (fcos.py:543-578): the checkpoint's own cls_logits conv, any kernel size the library
        packed (1x1 / 3x3)."""
        self.engine.head_pretrained()
        return self.engine.decode(out_sizes)


@CODE_GENERATOR_REGISTRY.register()
class CodeGenerator(HipComponent):
    """CodeGeneratorHead: forward_roi_align on the current support pyramid (code_generator.py:924-1002) and, with cls_norm=True,
    forward_normalize_code over a list of codes (:832-897)."""

    eval_shot = None
    engine_generator = "CodeGenerator"  # which in-library generator (sylph_config.cg_type) this component drives

    def __init__(self, cfg, feature_channels=256, feature_levels=5, strides=None):
        assert feature_channels == 256, "Each level must have the same channel!"
        self.has_scale = len(cfg.MODEL.META_LEARN.CODE_GENERATOR.get("SCALE_LAYER", [])) != 0  # -> "cls_weight_norm" outputs

    def __call__(self, boxes: Optional[torch.Tensor] = None, cls_norm: bool = False, class_codes=None, weight_norm=None):
        if cls_norm:
            return self.engine.normalize_codes(class_codes, weight_norm)
        code = self.engine.codegen(boxes)
        out = {"cls_conv": code[:256].reshape(1, 256, 1, 1), "cls_bias": code[256:257].reshape(1, 1, 1, 1)}
        if self.has_scale:  # code_generator.py:987-999
            out["cls_weight_norm"] = self.engine.codegen_weight_norm(1).reshape(1, 1, 1, 1)
        return out

    def forward_classes(self, boxes: torch.Tensor, shots: int):
        """Several classes of `shots` support boxes each in the current batch -> one code dict per class (the launches are
        shared, the per-class arithmetic is that of __call__)."""
        codes = self.engine.codegen_classes(boxes, shots)
        outs = [{"cls_conv": c[:256].reshape(1, 256, 1, 1), "cls_bias": c[256:257].reshape(1, 1, 1, 1)} for c in codes]
        if self.has_scale:
            wn = self.engine.codegen_weight_norm(len(outs))
            for o, w in zip(outs, wn):
                o["cls_weight_norm"] = w.reshape(1, 1, 1, 1)
        return outs


@CODE_GENERATOR_REGISTRY.register()
class ROIEncoder(HipComponent):
    """ROIEncoder.forward at inference (roi_encoder.py:146-204): one class of EVAL_SHOT support boxes -> its code."""

    engine_generator = "ROIEncoder"

    def __init__(self, cfg, feature_channels=256, feature_levels=5, strides=None):
        assert feature_channels == 256, "Each level must have the same channel!"
        self.eval_shot = int(cfg.MODEL.META_LEARN.EVAL_SHOT)

    def __call__(self, boxes: torch.Tensor):  # no cls_norm / class_codes keywords, like the reference
        code = self.engine.codegen(boxes)
        return {"cls_conv": code[:256].reshape(1, 256, 1, 1), "cls_bias": code[256:257].reshape(1)}

    def forward_classes(self, boxes: torch.Tensor, shots: int):
        """Several classes of EVAL_SHOT support boxes each in the current batch -> one code dict per class.  The launches are shared;
        a class's tokens never meet another class's (the reference's encoder sees one class per call at inference: a length-1
        sequence on its attention axis, roi_encoder.py:184-186), so the codes are those of __call__ per class."""
        assert shots == self.eval_shot, f"{shots} support images per class, EVAL_SHOT is {self.eval_shot}"
        codes = self.engine.codegen_classes(boxes, shots)
        return [{"cls_conv": c[:256].reshape(1, 256, 1, 1), "cls_bias": c[256:257].reshape(1)} for c in codes]


def build_code_generator(cfg, feature_channels, feature_levels, strides):
    """sylph/modeling/code_generator/build.py:30-39."""
    name = cfg.MODEL.META_LEARN.CODE_GENERATOR.NAME
    return CODE_GENERATOR_REGISTRY.get(name)(cfg, feature_channels, feature_levels, strides)


@META_ARCH_REGISTRY.register()
class MetaOneStageDetector(nn.Module):
    """Four inference forward types (meta_one_stage_detector.py:415-455):
      run_type None                         -> forward_base_detector for a non-episodic model (plain FCOS with the checkpoint's
                                               cls_logits); NotImplementedError for an episodic model (as the reference)
      "meta_learn_test_support"             -> forward_class_code
      "meta_learn_normalize_code"           -> normalize_class_code
      "meta_learn_test_instance"            -> forward_instances (class_code None: evaluation with the pretrained cls_logits)
    Training is out of scope: calling the model in training mode raises NotImplementedError."""

    def __init__(self, cfg, dtype: Optional[str] = None, device_index: Optional[int] = None):
        super().__init__()
        self.cfg = cfg
        self.episodic_learning = bool(cfg.MODEL.META_LEARN.EPISODIC_LEARNING)
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg)
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg)
        self.code_generator = (build_code_generator(cfg, 256, len(cfg.MODEL.FCOS.IN_FEATURES),
                                                    cfg.MODEL.FCOS.FPN_STRIDES) if self.episodic_learning else None)
        if self.episodic_learning:
            assert self.code_generator is not None
        self.in_features = list(cfg.MODEL.FCOS.IN_FEATURES)
        if device_index is None:
            dev = torch.device(cfg.MODEL.DEVICE)
            device_index = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        dtype = dtype or str(cfg.MODEL.get("COMPUTE_DTYPE", "bf16"))
        self.engine = Engine(cfg, dtype=dtype, device=device_index)
        for comp in (self.backbone, self.proposal_generator, self.code_generator):
            if hasattr(comp, "bind"):
                comp.bind(self.engine)
        self.register_buffer("pixel_mean", torch.tensor(list(cfg.MODEL.PIXEL_MEAN), dtype=torch.float32,
                                                        device=self.engine.device).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.tensor(list(cfg.MODEL.PIXEL_STD), dtype=torch.float32,
                                                       device=self.engine.device).view(-1, 1, 1))
        self._weights_loaded = False
        self.train(True)  # nn.Module default; the runner / predictor call .eval()

    # ---- reference surface -------------------------------------------------------------------------
    @property
    def device(self):
        return self.pixel_mean.device

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts the reference checkpoint's ["model"] dict (SURVEY.md 8b key layout)."""
        self.engine.load_state_dict(state_dict)
        self._weights_loaded = True
        # the base detector's own classifier (fcos.py:418-427): kept for eval_with_pretrained_code (class_code=None)
        w = state_dict.get("proposal_generator.fcos_head.cls_logits.weight")
        b = state_dict.get("proposal_generator.fcos_head.cls_logits.bias")
        self._pretrained_cls_logits = (torch.as_tensor(w).float(), torch.as_tensor(b).float()) if w is not None and b is not None else None
        return self

    def load_checkpoint(self, path: str):
        """DetectionCheckpointer.load (sylph/predictor.py:87-88): model_final.pth, detectron2 model-zoo .pkl or the MSRA
        R-50.pkl / R-101.pkl backbones the yamls name (sylph_amd.checkpoint maps them onto the reference keys)."""
        from .checkpoint import load_checkpoint_file
        return self.load_state_dict(load_checkpoint_file(path))

    def forward(self, batched_inputs, class_code=None, run_type=None):
        if self.training:
            raise NotImplementedError("training is out of scope of the MI355X inference path; call model.eval()")
        if run_type is None:
            # MetaProposalNetwork.forward (meta_one_stage_detector.py:120-141): a non-episodic model is a plain base detector
            if not self.episodic_learning:
                return self.forward_base_detector(batched_inputs)
            raise NotImplementedError(
                "Episodic learning inferrence for image and features is not supported in forward.")
        if run_type == "meta_learn_test_support":
            return self.forward_class_code(batched_inputs)
        if run_type == "meta_learn_normalize_code":
            return self.normalize_class_code(class_code)
        if run_type == "meta_learn_test_instance":
            return self.forward_instances(batched_inputs, class_code)
        raise NotImplementedError(f"not support this forward type: {run_type}, class_code: {class_code}")

    # ---- support path --------------------------------------------------------------------------------
    def forward_class_code(self, batched_inputs: List[Dict[str, Any]]) -> Dict[str, torch.Tensor]:
        """meta_one_stage_detector.py:229-254: ONE class, its S support records."""
        assert not self.training, "Not for training"
        assert len(batched_inputs) == 1, f"batched_inputs has length: {len(batched_inputs)}"
        records = [rec for x in batched_inputs for rec in x["support_set"]]
        boxes = []
        for rec in records:
            gt = rec["instances"].gt_boxes.tensor
            if len(gt) == 0:  # select_a_mask (code_generator/utils.py:35-38)
                logger.info("Run into empty box, use zero masks")
                raise ValueError
            if len(gt) > 1:
                # the reference draws np.random.choice here (global RNG, utils.py:41); deterministic
                # only when one box is given -- we require the caller to have selected it
                import numpy as np
                gt = gt[np.random.choice(range(len(gt)), 1)]
            boxes.append(gt.reshape(1, 4))
        num_shots = getattr(self.code_generator, "eval_shot", None)
        if num_shots:
            # roi_encoder.py:156-166: num_shots = EVAL_SHOT in eval, batch = total / num_shots (one class here)
            assert len(records) % num_shots == 0, f"{len(records)} % {num_shots}"
            assert len(records) // num_shots == 1, "one class per call at inference"
        self.backbone(images=[rec["image"] for rec in records])
        return self.code_generator(torch.cat(boxes, dim=0))

    def forward_class_codes(self, items: List[List[Dict[str, Any]]]) -> List[Dict[str, torch.Tensor]]:
        """Support-path throughput: the reference (and forward_class_code above) runs ONE class per call; here several classes
        share the backbone / code-generator launches of one batch (B = classes x shots).  `items`: loader items, each a list of
        length 1 as in forward_class_code.  Falls back to one call per class when the shot counts differ (or are not the
        ROIEncoder's EVAL_SHOT), a record carries more than one box, or the classes would be padded to different sizes when run
        alone (the padded size decides the activations along the right / bottom border, hence the code)."""
        assert not self.training, "Not for training"
        recs = [[rec for x in it for rec in x["support_set"]] for it in items]
        shots = len(recs[0]) if recs else 0
        from .evaluation import class_padded_hw
        d = getattr(self.backbone, "size_divisibility", 32)
        batchable = (len(items) > 1 and hasattr(self.code_generator, "forward_classes") and shots > 0
                     and getattr(self.code_generator, "eval_shot", None) in (None, shots)
                     and len({class_padded_hw(r, d) for r in recs if r}) == 1  # every class padded as in its own one-class call
                     and all(len(it) == 1 and len(r) == shots for it, r in zip(items, recs))
                     and all(len(rec["instances"].gt_boxes.tensor) == 1 for r in recs for rec in r))
        if not batchable:
            return [self.forward_class_code(it) for it in items]
        self.backbone(images=[rec["image"] for r in recs for rec in r])
        boxes = torch.cat([rec["instances"].gt_boxes.tensor.reshape(1, 4) for r in recs for rec in r], dim=0)
        return self.code_generator.forward_classes(boxes, shots)

    def normalize_class_code(self, codes: List[Dict]):
        """code_generator.py:877-897 via meta_one_stage_detector.py:256-259 (mutates the list)."""
        assert self.episodic_learning
        assert not self.training
        if isinstance(self.code_generator, ROIEncoder):
            # the reference fails the same way: ROIEncoder.forward() takes no cls_norm/class_codes
            # (meta_one_stage_detector.py:259 -> roi_encoder.py:146)
            raise TypeError("ROIEncoder.forward() got an unexpected keyword argument 'cls_norm'")
        assert codes is not None
        if len(codes) == 0:
            return codes
        rows, wns = [], []
        for code in codes:
            assert "class_code" in code, "class_code is not in code"
            assert "cls_conv" in code["class_code"], "class_conv is not in class_code"
            cc = code["class_code"]
            assert cc["cls_conv"].ndim == 4
            assert cc["cls_bias"].numel() == 1, "predicted bias should only have batch size 1"
            rows.append(torch.cat([cc["cls_conv"].reshape(-1).float(), cc["cls_bias"].reshape(-1).float()]))
            if "cls_weight_norm" in cc:  # x cls_weight_norm after the L2 normalisation (code_generator.py:838-840)
                wns.append(cc["cls_weight_norm"].reshape(-1).float())
        assert len(wns) in (0, len(rows)), "cls_weight_norm must be present for all classes or none"
        packed = torch.stack(rows).to(self.device).contiguous()
        out = self.code_generator(cls_norm=True, class_codes=packed, weight_norm=torch.cat(wns) if wns else None)
        for i, code in enumerate(codes):
            code["class_code"]["cls_conv"] = out[i, :256].reshape(1, 256, 1, 1)
            code["class_code"]["cls_bias"] = out[i, 256:257].reshape(1)
        return codes

    # ---- query path ----------------------------------------------------------------------------------
    def forward_base_detector(self, batched_inputs: List[Dict[str, Any]]):
        """meta_one_stage_detector.py:298-323 (inference branch) + the {"proposals"} -> {"instances"} renaming of
        MetaOneStageDetector.forward (:436-441): FCOS with the checkpoint's own cls_logits."""
        assert not self.episodic_learning
        return self._detect(batched_inputs, None)

    def forward_instances(self, batched_inputs: List[Dict[str, Any]], class_codes: Dict[str, torch.Tensor]):
        """meta_one_stage_detector.py:261-296 -> [{"instances": Instances}] at input["height"/"width"] scale."""
        assert self.episodic_learning
        return self._detect(batched_inputs, class_codes)

    def _detect(self, batched_inputs: List[Dict[str, Any]], class_codes):
        assert not self.training, "Not for training"
        pretrained = class_codes is None
        if class_codes is None:
            # MetaFCOSHead.forward with support_set_per_class_code=None -> forward_base_train (fcos.py:543-578):
            # logits = self.cls_logits(cls_tower), the pretrained base-class classifier.  A 1x1 cls_logits conv (the
            # CLS_LOGITS_KERNEL_SIZE of the Meta-FCOS recipes) is exactly the class-conditional conv with fixed codes.
            pre = getattr(self, "_pretrained_cls_logits", None)
            if pre is None:
                raise ValueError("class_code is None and the checkpoint has no proposal_generator.fcos_head.cls_logits weights")
            if pre[0].dim() != 4 or tuple(pre[0].shape[2:]) not in ((1, 1), (3, 3)):
                raise NotImplementedError(f"pretrained cls_logits with kernel {tuple(pre[0].shape[2:])}: only 1x1 and 3x3 "
                                          "(MODEL.FCOS.CLS_LOGITS_KERNEL_SIZE) are supported")
            if self.episodic_learning and not bool(self.cfg.MODEL.META_LEARN.CODE_GENERATOR.USE_BIAS):
                raise NotImplementedError("pretrained cls_logits needs the bias path of the class-conditional conv (USE_BIAS)")
            class_codes = {"cls_conv": pre[0].to(self.device), "cls_bias": pre[1].to(self.device)}
        w, b = class_codes["cls_conv"], class_codes.get("cls_bias")
        assert w.dim() == 4, f"Weight has dimension: {w.dim()}"
        assert w.size(1) == 256
        conv3 = pretrained and tuple(w.shape[2:]) == (3, 3)  # a 3x3 cls_logits is a real conv, not a class-conditional 1x1
        if all("image_u8" in x for x in batched_inputs):
            # fused input pipeline: the original uint8 HWC image + its ResizeShortestEdge target; resize,
            # BGR conversion, normalisation and padding run in one HIP kernel (sylph_preprocess_u8)
            sizes = self.backbone(images_u8=[x["image_u8"] for x in batched_inputs], resize_hw=[x["resize_hw"] for x in batched_inputs],
                                  rgb_input=str(batched_inputs[0].get("input_format", "BGR")) == "RGB")
        else:
            sizes = self.backbone(images=[x["image"] for x in batched_inputs])
        out_sizes = [(int(x.get("height", s[0])), int(x.get("width", s[1]))) for x, s in zip(batched_inputs, sizes)]
        # the checkpoint's own cls_logits are a plain conv: no CondConvBlock Scale on a ROIEncoder model
        if conv3:
            dets = self.proposal_generator.forward_pretrained(out_sizes)
        else:
            dets = self.proposal_generator(w, b, out_sizes, raw=True) if pretrained else self.proposal_generator(w, b, out_sizes)
        results = []
        for d, osz in zip(dets, out_sizes):
            r = Instances(osz)
            r.pred_boxes = Boxes(d["pred_boxes"])
            r.scores = d["scores"]
            r.pred_classes = d["pred_classes"]
            r.locations = d["locations"]
            r.fpn_levels = d["fpn_levels"]
            results.append({"instances": r})
        return results


def build_model(cfg, dtype: Optional[str] = None) -> nn.Module:
    """d2go runner.build_model equivalent for this path: META_ARCH_REGISTRY lookup by yaml name."""
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg, dtype=dtype)
