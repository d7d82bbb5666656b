"""Oracle: image preprocessing + ResNet-FPN backbone (fp32, torch CPU).  TEST INFRASTRUCTURE.

The arithmetic here lives in third-party packages that are NOT vendored under /root/reference
(detectron2@main ``modeling/backbone/{resnet,fpn}.py``, ``layers/batch_norm.py``,
``structures/image_list.py``; AdelaiDet@master ``adet/modeling/backbone/fpn.py``), pinned only
to a branch (requirements.txt:5-6).  It is restated from the published definitions and anchored
on the reference's call sites:
  * sylph/modeling/meta_arch/meta_one_stage_detector.py:60-65,174-178 (normalise + ImageList)
  * sylph/modeling/meta_arch/meta_one_stage_detector.py:75,101-115,181,273 (build_backbone,
    FrozenBN conversion, backbone call)
  * configs/COCO-Detection/Meta-FCOS/Base-FCOS.yaml:3-11 (build_fcos_resnet_fpn_backbone,
    res3..res5 -> FPN), sylph/runner/adet_configs.py:39 (TOP_LEVELS 2 -> P6,P7 from p5)
Parity for this file is UNPINNED by the reference (no numeric test exists there).  Third-party pins: the ResNet part (stem,
max-pool, bottleneck blocks incl. STRIDE_IN_1X1, projection shortcuts, FrozenBN) agrees with Hugging Face's independent
`transformers.ResNetModel(downsample_in_bottleneck=True)` to 2e-5 on res2..res5, R-50 and R-101, ragged sizes
(tests/test_oracle_vs_hf_resnet.py); the FPN part and P6/P7 by float64 known answers (tests/golden/gen_known_answers.py).
"""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

# detectron2 defaults (configs never override them; Base-FCOS.yaml:15 is commented out)
PIXEL_MEAN = (103.530, 116.280, 123.675)
PIXEL_STD = (1.0, 1.0, 1.0)
SIZE_DIVISIBILITY = 32
BN_EPS = 1e-5
STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def preprocess(images: Sequence[torch.Tensor], pixel_mean=PIXEL_MEAN, pixel_std=PIXEL_STD,
               size_divisibility: int = SIZE_DIVISIBILITY) -> Tuple[torch.Tensor, List[Tuple[int, int]]]:
    """(x - mean) / std per image, zero-pad (top-left anchored) to the batch max size rounded
    up to ``size_divisibility``.  meta_one_stage_detector.py:174-178; d2 ImageList.from_tensors."""
    mean = torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1)
    normed = [(x.float() - mean) / std for x in images]
    sizes = [(int(x.shape[-2]), int(x.shape[-1])) for x in normed]
    mh = max(s[0] for s in sizes)
    mw = max(s[1] for s in sizes)
    d = size_divisibility
    if d > 1:
        mh = (mh + d - 1) // d * d
        mw = (mw + d - 1) // d * d
    out = torch.zeros(len(normed), normed[0].shape[0], mh, mw, dtype=torch.float32)
    for i, x in enumerate(normed):
        out[i, :, : x.shape[-2], : x.shape[-1]] = x
    return out, sizes


def frozen_bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """detectron2 FrozenBatchNorm2d (eval): x * scale + shift, eps 1e-5."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w * (rv + BN_EPS).rsqrt()
    shift = b - rm * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def bn_scale_shift(sd, prefix):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w * (rv + BN_EPS).rsqrt()
    return scale, b - rm * scale


def _conv_bn(x, sd, name, stride=1, padding=0, relu=False):
    y = F.conv2d(x, sd[name + ".weight"], None, stride=stride, padding=padding)
    y = frozen_bn(y, sd, name + ".norm")
    return F.relu(y) if relu else y


def bottleneck(x, sd, prefix, stride, has_shortcut, stride_in_1x1=True):
    """detectron2 BottleneckBlock: 1x1 -> 3x3 -> 1x1 (+FrozenBN each), residual, ReLU.
    STRIDE_IN_1X1 is the d2 default (True): the stride sits on conv1 / shortcut."""
    s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
    out = _conv_bn(x, sd, prefix + ".conv1", stride=s1, relu=True)
    out = _conv_bn(out, sd, prefix + ".conv2", stride=s3, padding=1, relu=True)
    out = _conv_bn(out, sd, prefix + ".conv3")
    sc = _conv_bn(x, sd, prefix + ".shortcut", stride=stride) if has_shortcut else x
    return F.relu(out + sc)


def resnet(x: torch.Tensor, sd: Dict[str, torch.Tensor], depth: int = 50,
           prefix: str = "backbone.bottom_up") -> Dict[str, torch.Tensor]:
    """detectron2 ResNet bottom-up: BasicStem (7x7 s2 + FrozenBN + ReLU + maxpool 3x3 s2 p1),
    res2..res5.  Returns res3, res4, res5 (Base-FCOS.yaml:5-6 OUT_FEATURES)."""
    x = _conv_bn(x, sd, prefix + ".stem.conv1", stride=2, padding=3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for si, nblocks in enumerate(STAGE_BLOCKS[depth]):
        stage = si + 2
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and stage > 2) else 1
            x = bottleneck(x, sd, f"{prefix}.res{stage}.{bi}", stride, has_shortcut=(bi == 0))
        outs[f"res{stage}"] = x
    return outs


def fpn(feats: Dict[str, torch.Tensor], sd: Dict[str, torch.Tensor],
        prefix: str = "backbone") -> Dict[str, torch.Tensor]:
    """detectron2 FPN over res3..res5 (fuse 'sum', nearest x2 top-down) + AdelaiDet
    LastLevelP6P7(in_feature='p5'): p6 = conv3x3 s2 (p5); p7 = conv3x3 s2 (relu(p6))."""
    def conv(x, name, stride=1, padding=0):
        return F.conv2d(x, sd[f"{prefix}.{name}.weight"], sd[f"{prefix}.{name}.bias"],
                        stride=stride, padding=padding)
    prev = conv(feats["res5"], "fpn_lateral5")
    out = {"p5": conv(prev, "fpn_output5", padding=1)}
    for stage in (4, 3):
        top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        lat = conv(feats[f"res{stage}"], f"fpn_lateral{stage}")
        prev = lat + top_down
        out[f"p{stage}"] = conv(prev, f"fpn_output{stage}", padding=1)
    p6 = conv(out["p5"], "top_block.p6", stride=2, padding=1)
    p7 = conv(F.relu(p6), "top_block.p7", stride=2, padding=1)
    out["p6"], out["p7"] = p6, p7
    return out


def backbone_fpn(images: torch.Tensor, sd, depth: int = 50) -> List[torch.Tensor]:
    """images (B,3,H,W) already normalised/padded -> [p3,p4,p5,p6,p7] (B,256,h,w)."""
    f = fpn(resnet(images, sd, depth), sd)
    return [f[k] for k in ("p3", "p4", "p5", "p6", "p7")]
