"""Oracle: Meta-FCOS head (towers, box/ctrness/iou predictors, class-conditional classifier).
TEST INFRASTRUCTURE.  fp32 torch CPU.

Follows (paths relative to /root/reference):
  * sylph/modeling/meta_fcos/fcos.py:72-122   _build_tower_module: N x [conv3x3 C->C + bias,
    GroupNorm(32, C), ReLU] for "cls" and "bbox" (NUM_SHARE_CONVS = 0 -> identity share tower)
  * sylph/modeling/meta_fcos/fcos.py:382-484  layer shapes (bbox_pred 4, ctrness 1, iou_overlap 1,
    per-level Scale)
  * sylph/modeling/meta_fcos/fcos.py:582-667  MetaFCOSHead.forward (episodic branch)
  * sylph/modeling/meta_fcos/head_utils.py:23-29   Scale
  * sylph/modeling/meta_fcos/head_utils.py:39-81   CondConvBasic
  * sylph/modeling/meta_fcos/head_utils.py:121-162 CondConvBlock (ROIEncoder variant)
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

GN_GROUPS = 32
GN_EPS = 1e-5
HEAD_PREFIX = "proposal_generator.fcos_head"


def tower(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, num_convs: int = 4, norm: str = "GN") -> torch.Tensor:
    """fcos.py:72-122.  nn.Sequential indices with MODEL.FCOS.NORM "GN" / "NaiveGN" (adet NaiveGroupNorm: the same arithmetic): conv 3i,
    GN 3i+1, ReLU 3i+2; with "none": conv 2i, ReLU 2i+1."""
    step = 2 if norm in ("none", "", None) else 3
    for i in range(num_convs):
        x = F.conv2d(x, sd[f"{prefix}.{step * i}.weight"], sd[f"{prefix}.{step * i}.bias"], padding=1)
        if step == 3:
            x = F.group_norm(x, GN_GROUPS, sd[f"{prefix}.{3 * i + 1}.weight"],
                             sd[f"{prefix}.{3 * i + 1}.bias"], eps=GN_EPS)
        x = F.relu(x)
    return x


def cond_conv_basic(feature: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                    padding: int = 0, stride: int = 1, use_bias: bool = True) -> torch.Tensor:
    """head_utils.py:60-81 (same asserts -> AssertionError)."""
    assert feature.size(1) == weight.size(1)
    assert feature.dim() == 4, f"Feature has dimension: {feature.dim()}"
    assert weight.dim() == 4, f"Weight has dimension: {weight.dim()}"
    return F.conv2d(feature, weight, bias=bias if use_bias else None, stride=stride, padding=padding)


def cond_conv_block(feature, weight, bias=None, scales: Optional[List[float]] = None, padding: int = 0):
    """head_utils.py:140-162.  weight (N, 256*k, 1, 1); Scale init 1/k each.  Note the
    reference reuses index i (not i+1) for chunks past the first (head_utils.py:157-161)."""
    assert len(weight.shape) == 4, f"weight has wrong shape, {weight.shape}"
    k = weight.size(1) // 256
    assert weight.size(1) == 256 * k and k >= 1
    if scales is None:
        scales = [1.0 / k] * k
    out = scales[0] * F.conv2d(feature, weight[:, 0:256], bias, padding=padding)
    for i in range(k - 1):
        s = 256 * (i + 1)
        out = out + scales[i] * F.conv2d(feature, weight[:, s:s + 256], bias, padding=padding)
    return out


def fcos_head(features: List[torch.Tensor], sd: Dict[str, torch.Tensor], class_codes: Dict[str, torch.Tensor],
              num_cls_convs: int = 4, num_box_convs: int = 4, use_scale: bool = True,
              use_bias: bool = True, cond_block: bool = False, prefix: str = HEAD_PREFIX,
              cond_scales: Optional[List[float]] = None, num_share_convs: int = 0, norm: str = "GN"):
    """fcos.py:582-667 with support_set_per_class_code given.  Returns per-level lists
    (logits (B,N,h,w), reg (B,4,h,w) = relu(scale_l * bbox_pred), ctrness (B,1,h,w), iou (B,1,h,w))."""
    w = class_codes["cls_conv"]
    b = class_codes["cls_bias"]
    logits, regs, ctrs, ious = [], [], [], []
    for level, feat in enumerate(features):
        feat = tower(feat, sd, f"{prefix}.share_tower", num_share_convs, norm)  # fcos.py:626 (identity for NUM_SHARE_CONVS = 0)
        cls_t = tower(feat, sd, f"{prefix}.cls_tower", num_cls_convs, norm)
        box_t = tower(feat, sd, f"{prefix}.bbox_tower", num_box_convs, norm)
        if cond_block:
            logit = cond_conv_block(cls_t, w, b, scales=cond_scales)  # Scale parameters of the checkpoint (head_utils.py:131-136)
        else:
            logit = cond_conv_basic(cls_t, w, b, padding=0, use_bias=use_bias)
        reg = F.conv2d(box_t, sd[f"{prefix}.bbox_pred.weight"], sd[f"{prefix}.bbox_pred.bias"], padding=1)
        if use_scale:
            reg = reg * sd[f"{prefix}.scales.{level}.scale"]
        regs.append(F.relu(reg))
        logits.append(logit)
        ctrs.append(F.conv2d(box_t, sd[f"{prefix}.ctrness.weight"], sd[f"{prefix}.ctrness.bias"], padding=1))
        ious.append(F.conv2d(box_t, sd[f"{prefix}.iou_overlap.weight"], sd[f"{prefix}.iou_overlap.bias"], padding=1))
    return logits, regs, ctrs, ious
