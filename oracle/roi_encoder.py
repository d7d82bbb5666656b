"""Oracle: the ROIEncoder code generator (LVIS variant).  TEST INFRASTRUCTURE.  fp32 torch CPU.

Follows (paths relative to /root/reference):
  * sylph/modeling/code_generator/roi_encoder.py:146-204   ROIEncoder.forward (eval: num_shots = EVAL_SHOT)
  * sylph/modeling/code_generator/roi_encoder.py:26-79     Tokenizer (conv3x3 [no bias with norm] + GN + ReLU, Flatten,
                                                           FC + ReLU)
  * sylph/modeling/code_generator/roi_encoder.py:82-115    HyperNetworkHead
  * sylph/modeling/code_generator/utils.py:70-103          MS_CAM
  * sylph/modeling/code_generator/utils.py:106-165         FeatureFusionModuleV2 (ROI pool -> conv+GN+ReLU -> context gate)
  * torch.nn.TransformerEncoder (post-norm, ReLU, batch_first=False): the reference feeds (bs, shots, C), so
    attention runs over the CLASS axis bs and is batched over shots (roi_encoder.py:184-186)
"""
import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from .roi_align import roi_pooler

P = "code_generator"


def ms_cam(x: torch.Tensor, context: torch.Tensor, sd, prefix: str) -> torch.Tensor:
    def branch(t, name, idx):
        t = F.conv2d(t, sd[f"{prefix}.{name}.{idx[0]}.weight"], sd[f"{prefix}.{name}.{idx[0]}.bias"])
        t = F.relu(F.group_norm(t, 32, sd[f"{prefix}.{name}.{idx[1]}.weight"], sd[f"{prefix}.{name}.{idx[1]}.bias"], eps=1e-5))
        t = F.conv2d(t, sd[f"{prefix}.{name}.{idx[2]}.weight"], sd[f"{prefix}.{name}.{idx[2]}.bias"])
        return F.group_norm(t, 32, sd[f"{prefix}.{name}.{idx[3]}.weight"], sd[f"{prefix}.{name}.{idx[3]}.bias"], eps=1e-5)
    local = branch(context, "local_att", (0, 1, 3, 4))
    glob = branch(F.adaptive_avg_pool2d(context, 1), "global_att", (1, 2, 4, 5))
    return x * torch.sigmoid(local + glob)


def encoder_layer(x: torch.Tensor, sd, q: str, nhead: int) -> torch.Tensor:
    """x (L, N, E): post-norm TransformerEncoderLayer, eval (dropout off)."""
    L, N, E = x.shape
    qkv = F.linear(x, sd[f"{q}.self_attn.in_proj_weight"], sd[f"{q}.self_attn.in_proj_bias"])
    qh, kh, vh = [t.reshape(L, N * nhead, E // nhead).transpose(0, 1) for t in qkv.chunk(3, dim=-1)]
    att = torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(E // nhead), dim=-1) @ vh
    att = att.transpose(0, 1).reshape(L, N, E)
    sa = F.linear(att, sd[f"{q}.self_attn.out_proj.weight"], sd[f"{q}.self_attn.out_proj.bias"])
    x = F.layer_norm(x + sa, (E,), sd[f"{q}.norm1.weight"], sd[f"{q}.norm1.bias"], eps=1e-5)
    ff = F.linear(F.relu(F.linear(x, sd[f"{q}.linear1.weight"], sd[f"{q}.linear1.bias"])),
                  sd[f"{q}.linear2.weight"], sd[f"{q}.linear2.bias"])
    return F.layer_norm(x + ff, (E,), sd[f"{q}.norm2.weight"], sd[f"{q}.norm2.bias"], eps=1e-5)


def roi_encoder(features: List[torch.Tensor], boxes: torch.Tensor, sd, num_shots: int, strides=(8, 16, 32, 64, 128),
                tok_convs: int = 2, tok_fcs: int = 2, layers: int = 2, nhead: int = 8, head_fcs: int = 2,
                prior_prob: float = 0.01) -> Dict[str, torch.Tensor]:
    total = features[0].shape[0]
    assert total % num_shots == 0, f"{total} % {num_shots}"
    pooled = roi_pooler(features, boxes, strides, out_size=7)
    bp = f"{P}.box_pooler"
    pooled = F.conv2d(pooled, sd[f"{bp}.conv.0.weight"], sd[f"{bp}.conv.0.bias"], padding=1)
    pooled = F.relu(F.group_norm(pooled, 32, sd[f"{bp}.conv.1.weight"], sd[f"{bp}.conv.1.bias"], eps=1e-5))
    context = torch.stack([F.adaptive_avg_pool2d(f, (7, 7)) for f in features]).mean(dim=0)
    x = ms_cam(pooled, context, sd, f"{bp}.context_attention_module")
    for k in range(tok_convs):
        x = F.conv2d(x, sd[f"{P}.tokenizer.conv{k + 1}.weight"], None, padding=1)
        x = F.relu(F.group_norm(x, 32, sd[f"{P}.tokenizer.conv{k + 1}.norm.weight"],
                                sd[f"{P}.tokenizer.conv{k + 1}.norm.bias"], eps=1e-5))
    x = x.flatten(1)
    for k in range(tok_fcs):
        x = F.relu(F.linear(x, sd[f"{P}.tokenizer.fc{k + 1}.weight"], sd[f"{P}.tokenizer.fc{k + 1}.bias"]))
    tokens = x.view(-1, num_shots, x.shape[-1])      # (bs, shots, C) fed as (L=bs, N=shots, E)
    for l in range(layers):
        tokens = encoder_layer(tokens, sd, f"{P}.transformer_encoder.layers.{l}", nhead)
    cls_tok = tokens.mean(1)

    def head(t, name):
        for i in range(head_fcs):
            t = F.linear(t, sd[f"{P}.{name}.fc{i + 1}.weight"], sd[f"{P}.{name}.fc{i + 1}.bias"])
            if i < head_fcs - 1:
                t = F.relu(t)
        return t
    w = head(cls_tok, "weight_head")
    b = head(cls_tok, "bias_head")
    bias_value = -math.log((1 - prior_prob) / prior_prob)
    return {"cls_conv": w.view(w.size(0), w.size(1), 1, 1), "cls_bias": (bias_value + b).view(-1)}
