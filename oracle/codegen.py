"""Oracle: Sylph hypernetwork ("code generator") head and class-code normalisation.
TEST INFRASTRUCTURE.  fp32 torch CPU.

Follows (paths relative to /root/reference):
  * sylph/modeling/code_generator/code_generator.py:648-688  support_set_shared_tower
    (conv3x3 + bias, GroupNorm(32), ReLU per TOWER_LAYERS entry; Sequential indices 3i, 3i+1)
  * sylph/modeling/code_generator/code_generator.py:509-580  support_set_cls_conv (conv3x3
    256->OUT + global avg pool), support_set_cls_bias (conv3x3 256->1 [+ pool])
  * sylph/modeling/code_generator/code_generator.py:924-1002 forward_roi_align (eval branch)
  * sylph/modeling/code_generator/code_generator.py:766-829  process_weight / compute_code
  * sylph/modeling/code_generator/code_generator.py:832-897  normalize_code, process_bias,
    code_process_module, forward_normalize_code
  * sylph/modeling/code_generator/utils.py:51-67             GlobalAdaptiveAvgPool2d
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .roi_align import roi_pooler

CG_PREFIX = "code_generator.code_generator_head"
GN_EPS = 1e-5


def bias_prior(prior_prob: float = 0.01) -> float:
    """code_generator.py:422-425: -log((1-p)/p) (= -4.59512 for p = 0.01)."""
    return -math.log((1 - prior_prob) / prior_prob)


def shared_tower(x: torch.Tensor, sd, n_layers: int = 2, prefix: str = CG_PREFIX, spec=None) -> torch.Tensor:
    """code_generator.py:648-688.  spec = CODE_GENERATOR.TOWER_LAYERS entries [norm, act] (default n_layers x ["GN", "ReLU"]): norm
    "GN" / "" (build_fpn_norm), act "ReLU" / "Tanh" / ""; the nn.Sequential index advances per existing module."""
    spec = spec if spec is not None else [["GN", "ReLU"]] * n_layers
    idx = 0
    for norm, act in spec:
        x = F.conv2d(x, sd[f"{prefix}.support_set_shared_tower.{idx}.weight"],
                     sd[f"{prefix}.support_set_shared_tower.{idx}.bias"], padding=1)
        idx += 1
        if norm == "GN":
            x = F.group_norm(x, 32, sd[f"{prefix}.support_set_shared_tower.{idx}.weight"],
                             sd[f"{prefix}.support_set_shared_tower.{idx}.bias"], eps=GN_EPS)
            idx += 1
        elif norm not in ("", "none", None):
            raise NotImplementedError(norm)
        if act == "ReLU":
            x = F.relu(x); idx += 1
        elif act == "Tanh":
            x = torch.tanh(x); idx += 1
    return x


def code_from_roi_features(roi: torch.Tensor, sd, n_tower_layers: int = 2, bias_l2_norm: bool = False,
                           has_bias_layer: bool = True, has_weight_layer: bool = False, has_scale_layer: bool = False,
                           prefix: str = CG_PREFIX, tower_spec=None) -> Dict[str, torch.Tensor]:
    """roi (S,256,7,7): ALL S shots belong to one class (eval: num_shot = batch,
    code_generator.py:788-792).  Returns un-normalised cls_conv (1,OUT,1,1), cls_bias (1,1,1,1) and, with a SCALE_LAYER,
    cls_weight_norm (1,1,1,1).  WEIGHT_LAYER: softmax over the shots of a pooled 1-channel head replaces the uniform
    shot weights (code_generator.py:583-613,766-777,969-979)."""
    f = shared_tower(roi, sd, n_tower_layers, prefix, tower_spec)
    conv_feat = F.conv2d(f, sd[f"{prefix}.support_set_cls_conv.0.weight"],
                         sd[f"{prefix}.support_set_cls_conv.0.bias"], padding=1)
    conv_feat = F.adaptive_avg_pool2d(conv_feat, (1, 1))
    S = roi.shape[0]
    w = torch.full((1, S, 1, 1, 1), 1.0 / S)  # code_generator.py:803-804 (uniform weights)
    if has_weight_layer:
        wl = F.adaptive_avg_pool2d(F.conv2d(f, sd[f"{prefix}.support_set_cls_weight.0.weight"],
                                            sd[f"{prefix}.support_set_cls_weight.0.bias"], padding=1), (1, 1))
        w = torch.softmax(wl.view(-1, S, 1, 1, 1), dim=1)  # process_weight (code_generator.py:766-777)
    cls_conv = (w * conv_feat.view(1, S, conv_feat.size(1), 1, 1)).sum(dim=1)
    cls_bias = torch.zeros(1, 1, 1, 1)
    if has_bias_layer:
        bias_feat = F.conv2d(f, sd[f"{prefix}.support_set_cls_bias.0.weight"],
                             sd[f"{prefix}.support_set_cls_bias.0.bias"], padding=1)
        if bias_l2_norm:  # code_generator.py:962-967
            shp = bias_feat.size()
            bias_feat = F.normalize(bias_feat.view(shp[0], shp[1], -1), p=2, dim=2).view(shp)
        bias_feat = F.adaptive_avg_pool2d(bias_feat, (1, 1))
        cls_bias = (w * bias_feat.view(1, S, 1, 1, 1)).sum(dim=1)
    out = {"cls_conv": cls_conv, "cls_bias": cls_bias}
    if has_scale_layer:  # code_generator.py:976-993
        sc = F.adaptive_avg_pool2d(F.conv2d(f, sd[f"{prefix}.support_set_cls_scale.0.weight"],
                                            sd[f"{prefix}.support_set_cls_scale.0.bias"], padding=1), (1, 1))
        out["cls_weight_norm"] = (w * sc.view(1, S, 1, 1, 1)).sum(dim=1)
    return out


def code_generator(features: List[torch.Tensor], boxes: torch.Tensor, sd, strides=(8, 16, 32, 64, 128),
                   **kw) -> Dict[str, torch.Tensor]:
    """code_generator.py:924-1002: ROI pool (one box per support image) then the head.  (ROI_BOX.FPN_MULTILEVEL_FEATURE cannot run
    in the reference: CodeGeneratorHead builds detectron2's single-level-assignment ROIPooler (:26,343), so :943 iterates over the
    batch dimension of ONE tensor and GroupNorm fails on the unbatched (256, 7, 7) slices.)"""
    roi = roi_pooler(features, boxes, strides, out_size=7)
    return code_from_roi_features(roi, sd, **kw)


def normalize_code(cls_conv: torch.Tensor, cls_bias: torch.Tensor, sd, post_norm: bool = True,
                   conv_l2_norm: bool = True, use_weight_scale: bool = True, prior_prob: float = 0.01,
                   cls_weight_norm: Optional[torch.Tensor] = None, prefix: str = CG_PREFIX):
    """code_generator.py:832-875 for one class: GN(32) -> L2 normalise over C -> (x weight_norm)
    -> x conv_scale; bias * bias_scale + prior.  Returns (cls_conv (1,C,1,1), cls_bias (1,))."""
    assert cls_conv.ndim == 4
    code = cls_conv
    if post_norm and code.size(1) % 32 == 0:
        code = F.group_norm(code, 32, sd[f"{prefix}.post_norm.weight"], sd[f"{prefix}.post_norm.bias"], eps=GN_EPS)
    if conv_l2_norm:
        code = F.normalize(code, p=2, dim=1)
    if cls_weight_norm is not None:
        code = code * cls_weight_norm
    if use_weight_scale and (conv_l2_norm or post_norm):
        code = code * sd[f"{prefix}.conv_scale.scale"]
    assert cls_bias.size(0) == 1, "predicted bias should only have batch size 1"
    bias = cls_bias.view(cls_bias.numel())
    if f"{prefix}.bias_scale.scale" in sd:
        bias = bias * sd[f"{prefix}.bias_scale.scale"]
    bias = bias + torch.tensor(bias_prior(prior_prob), dtype=torch.float32)
    return code, bias


def forward_normalize_code(codes: List[Dict], sd, **kw) -> List[Dict]:
    """code_generator.py:877-897: in-place over a list of {"class_code": {...}} records."""
    for code in codes:
        assert "class_code" in code, "class_code is not in code"
        assert "cls_conv" in code["class_code"], "class_conv is not in class_code"
        cc = code["class_code"]
        cc["cls_conv"], cc["cls_bias"] = normalize_code(cc["cls_conv"], cc["cls_bias"], sd,
                                                        cls_weight_norm=cc.get("cls_weight_norm"), **kw)
    return codes
