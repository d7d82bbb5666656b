"""Oracle, bf16-storage mode: the same layer graph as oracle/backbone.py + oracle/head.py, but every tensor the
HIP throughput path STORES as bf16 is rounded to bf16 at the same point (round-to-nearest-even, like
v_cvt_pk_bf16_f32), with fp32 accumulation and fp32 epilogue arithmetic in between.  TEST INFRASTRUCTURE
(tests/, smoke(), bench.py's parity leg only); torch CPU.

With identical operands a HIP kernel and this restatement then round the SAME fp32 value up to accumulation
order (~1e-6 relative), so per-kernel outputs agree bit for bit except where that value sits on a bf16
rounding boundary (a 1-ulp flip, 2^-8 relative): the bf16 production kernels can be pinned to ulps instead
of to a cosine.

Storage points restated here:
  * network input after (x - mean) / std                          -> bf16   (preprocess_kernel)
  * every conv epilogue: acc * bn_scale + bn_shift (one fma) [+ residual] [ReLU] -> bf16
    (conv_igemm / conv_pw / conv_hpipe / bottleneck64[p] / stem_pool), weights bf16, un-scaled
  * a block with a projection shortcut: conv3 and the shortcut are ONE GEMM over K = [t2 | x]; the two
    FrozenBN scales are folded into the weights in fp32 BEFORE the bf16 cast, the shifts are summed
    (api_weights.hip make_c3sc)
  * FPN lateral (+ nearest-2x top-down), output convs, P6, relu(P6), P7: bias epilogue -> bf16
  * FCOS towers: conv + bias -> bf16 (stored PRE-GroupNorm); GroupNorm statistics from the fp32 epilogue
    values (before rounding); the consumer applies x <- bf16(relu(fma(a, x, b))) to the stored bf16 values
  * class-conditional conv: bf16 normalised features x bf16 codes, fp32 logits + bias
  * bbox/ctrness/iou prediction conv: bf16 normalised features x bf16 weights, fp32, + bias, Scale, ReLU

Reference ops: the same as oracle/backbone.py and oracle/head.py (detectron2 ResNet/FPN at the call sites
sylph/modeling/meta_arch/meta_one_stage_detector.py:181,273; sylph/modeling/meta_fcos/fcos.py:72-122,582-667).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import backbone as _bb
from .head import GN_EPS, GN_GROUPS, HEAD_PREFIX


def r(t: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (nearest even) and widen back: the value a bf16 store + load yields."""
    return t.to(torch.bfloat16).to(torch.float32)


def fma(x: torch.Tensor, a, b) -> torch.Tensor:
    """fp32 fused multiply-add x * a + b (one rounding): the product of two fp32 values is exact in float64."""
    a = a if torch.is_tensor(a) else torch.tensor(a, dtype=torch.float32)
    b = b if torch.is_tensor(b) else torch.tensor(b, dtype=torch.float32)
    return (x.double() * a.double() + b.double()).float()


def _cv(v: Optional[torch.Tensor]):
    return v.view(1, -1, 1, 1) if v is not None else None


def conv_epilogue(x_bf: torch.Tensor, w: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                  stride: int = 1, padding: int = 0, relu: bool = False, res_bf: Optional[torch.Tensor] = None):
    """One conv launch: bf16 operands, fp32 accumulate, v = fma(acc, scale, shift) (+ residual) (ReLU).
    Returns (v fp32 before the store, r(v) as stored)."""
    acc = F.conv2d(x_bf, r(w), None, stride=stride, padding=padding)
    sc = _cv(scale) if scale is not None else 1.0
    sh = _cv(shift) if shift is not None else 0.0
    v = fma(acc, sc, sh)
    if res_bf is not None:
        v = v + res_bf
    if relu:
        v = F.relu(v)
    return v, r(v)


def preprocess(images: Sequence[torch.Tensor]):
    x, sizes = _bb.preprocess(images)
    return r(x), sizes


def stem_pool(x_bf, sd, prefix="backbone.bottom_up"):
    sc, sh = _bb.bn_scale_shift(sd, prefix + ".stem.conv1.norm")
    _, y = conv_epilogue(x_bf, sd[prefix + ".stem.conv1.weight"], sc, sh, stride=2, padding=3, relu=True)
    return F.max_pool2d(y, kernel_size=3, stride=2, padding=1)


def bottleneck_params(sd, prefix, has_shortcut):
    """-> (ws, scales, shifts) of conv1, conv2, conv3, shortcut (None for an identity block)."""
    names = ["conv1", "conv2", "conv3"] + (["shortcut"] if has_shortcut else [])
    ws = [sd[f"{prefix}.{n}.weight"] for n in names]
    ss = [_bb.bn_scale_shift(sd, f"{prefix}.{n}.norm") for n in names]
    return ws, [s[0] for s in ss], [s[1] for s in ss]


def bottleneck(x_bf, ws, scales, shifts, stride, stride_in_1x1=True):
    """One block as the HIP graph computes it (t1, t2 stored bf16; projection folded into conv3's GEMM)."""
    s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
    _, t1 = conv_epilogue(x_bf, ws[0], scales[0], shifts[0], stride=s1, relu=True)
    _, t2 = conv_epilogue(t1, ws[1], scales[1], shifts[1], stride=s3, padding=1, relu=True)
    if len(ws) > 3 and ws[3] is not None:
        w3f = r(ws[2] * scales[2].view(-1, 1, 1, 1))
        wsf = r(ws[3] * scales[3].view(-1, 1, 1, 1))
        acc = F.conv2d(t2, w3f) + F.conv2d(x_bf, wsf, stride=stride)
        v = F.relu(acc + (shifts[2] + shifts[3]).view(1, -1, 1, 1))
        return r(v)
    _, y = conv_epilogue(t2, ws[2], scales[2], shifts[2], relu=True, res_bf=x_bf)
    return y


def resnet(x_bf, sd, depth=50, prefix="backbone.bottom_up", start_stage=2, x_stage=None):
    """bf16-storage ResNet bottom-up.  start_stage > 2 with x_stage: start from a given res<start_stage - 1> output
    (stage-wise comparison against sylph_export_stage).  Returns {res2..res5}."""
    x = stem_pool(x_bf, sd, prefix) if start_stage == 2 else x_stage
    outs = {}
    for si, nblocks in enumerate(_bb.STAGE_BLOCKS[depth]):
        stage = si + 2
        if stage < start_stage:
            continue
        for bi in range(nblocks):
            stride = 2 if (bi == 0 and stage > 2) else 1
            ws, ss, hs = bottleneck_params(sd, f"{prefix}.res{stage}.{bi}", bi == 0)
            x = bottleneck(x, ws, ss, hs, stride)
        outs[f"res{stage}"] = x
    return outs


def fpn(feats: Dict[str, torch.Tensor], sd, prefix="backbone"):
    def conv(x, name, stride=1, padding=0, res=None):
        return conv_epilogue(x, sd[f"{prefix}.{name}.weight"], None, sd[f"{prefix}.{name}.bias"], stride=stride,
                             padding=padding, res_bf=res)[1]
    prev = conv(feats["res5"], "fpn_lateral5")
    out = {"p5": conv(prev, "fpn_output5", padding=1)}
    for stage in (4, 3):
        top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = conv(feats[f"res{stage}"], f"fpn_lateral{stage}", res=top_down)
        out[f"p{stage}"] = conv(prev, f"fpn_output{stage}", padding=1)
    p6 = conv(out["p5"], "top_block.p6", stride=2, padding=1)
    out["p6"] = p6
    out["p7"] = conv(F.relu(p6), "top_block.p7", stride=2, padding=1)
    return out


def backbone_fpn(x_bf, sd, depth=50) -> List[torch.Tensor]:
    f = fpn(resnet(x_bf, sd, depth), sd)
    return [f[k] for k in ("p3", "p4", "p5", "p6", "p7")]


# ---- head -------------------------------------------------------------------------------------------
def gn_coef(v: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """GroupNorm(32) statistics of the fp32 epilogue values v (B,256,h,w) -> (B,256,2) coefficients (a, b) with
    GN(x) = a * x + b (elementwise.hip gn_finalize_partials_kernel: float64 merge; gn_coef_kernel: fp32)."""
    B, C = v.shape[0], v.shape[1]
    g = v.double().reshape(B, GN_GROUPS, -1)
    mean = g.mean(dim=2)
    var = g.var(dim=2, unbiased=False)
    rstd = (1.0 / torch.sqrt(var + GN_EPS)).float().repeat_interleave(C // GN_GROUPS, dim=1)
    mean = mean.float().repeat_interleave(C // GN_GROUPS, dim=1)
    a = rstd * gamma.view(1, -1)
    b = beta.view(1, -1) - mean * a
    return torch.stack([a, b], dim=2)


def gn_apply(y_bf: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    """x <- bf16(relu(fma(a, y, b))): what the consumer of a tower layer feeds its MFMAs."""
    a = coef[:, :, 0].reshape(coef.shape[0], -1, 1, 1)
    b = coef[:, :, 1].reshape(coef.shape[0], -1, 1, 1)
    return r(F.relu(fma(y_bf, a, b)))


def tower_layer(x_bf, sd, prefix, i):
    """conv3x3 + bias of tower layer i -> (fp32 epilogue values, stored bf16 values, GroupNorm coefficients)."""
    v, y = conv_epilogue(x_bf, sd[f"{prefix}.{3 * i}.weight"], None, sd[f"{prefix}.{3 * i}.bias"], padding=1)
    return v, y, gn_coef(v, sd[f"{prefix}.{3 * i + 1}.weight"], sd[f"{prefix}.{3 * i + 1}.bias"])


def tower(x_bf, sd, prefix, num_convs=4):
    """-> normalised bf16 features of the last layer (the MFMA operand of the prediction passes)."""
    for i in range(num_convs):
        _, y, cf = tower_layer(x_bf, sd, prefix, i)
        x_bf = gn_apply(y, cf)
    return x_bf


def cls_logits(xn_bf, w, b):
    """Class-conditional 1x1 conv: bf16 codes (pack_codes_kernel), fp32 accumulate, + bias."""
    return F.conv2d(xn_bf, r(w), None) + (_cv(b) if b is not None else 0.0)


def predictions(xn_bf, sd, level, use_scale=True, prefix=HEAD_PREFIX):
    """-> reg (B,4,h,w) = relu(scale_l * (bbox_pred + bias)), ctrness, iou_overlap (fp32)."""
    reg = F.conv2d(xn_bf, r(sd[f"{prefix}.bbox_pred.weight"]), None, padding=1) + _cv(sd[f"{prefix}.bbox_pred.bias"])
    if use_scale:
        reg = reg * sd[f"{prefix}.scales.{level}.scale"]
    ctr = F.conv2d(xn_bf, r(sd[f"{prefix}.ctrness.weight"]), None, padding=1) + _cv(sd[f"{prefix}.ctrness.bias"])
    iou = F.conv2d(xn_bf, r(sd[f"{prefix}.iou_overlap.weight"]), None, padding=1) + _cv(sd[f"{prefix}.iou_overlap.bias"])
    return F.relu(reg), ctr, iou


def fcos_head(features_bf: List[torch.Tensor], sd, class_codes, num_convs=4, use_scale=True, prefix=HEAD_PREFIX):
    w, b = class_codes["cls_conv"], class_codes["cls_bias"]
    logits, regs, ctrs, ious = [], [], [], []
    for level, feat in enumerate(features_bf):
        ct = tower(feat, sd, f"{prefix}.cls_tower", num_convs)
        bt = tower(feat, sd, f"{prefix}.bbox_tower", num_convs)
        logits.append(cls_logits(ct, w, b))
        reg, ctr, iou = predictions(bt, sd, level, use_scale, prefix)
        regs.append(reg); ctrs.append(ctr); ious.append(iou)
    return logits, regs, ctrs, ious


def forward_instances(images, class_codes, sd, depth=50, post_nms_topk=100, **decode_kw):
    """bf16-storage twin of oracle.episode.forward_instances (decode itself is fp32 in both paths)."""
    from . import decode as _dec
    x, sizes = preprocess(images)
    feats = backbone_fpn(x, sd, depth)
    logits, regs, ctrs, ious = fcos_head(feats, sd, class_codes)
    props = _dec.predict_proposals(logits, regs, ctrs, ious, post_nms_topk=post_nms_topk, **decode_kw)
    return [_dec.detector_postprocess(p, sizes[i], sizes[i][0], sizes[i][1]) for i, p in enumerate(props)]


def ulp_report(got: torch.Tensor, want: torch.Tensor) -> Tuple[float, float]:
    """(fraction of elements that are not bit-identical, worst |difference| in units of the bf16 spacing at that
    magnitude).  Both tensors hold bf16-representable values."""
    diff = (got - want).abs()
    mag = torch.maximum(got.abs(), want.abs()).clamp_min(2.0 ** -126)
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)   # spacing of bf16 (8 significant bits) at that binade
    return float((diff > 0).float().mean()), float((diff / ulp).max())
