"""Seeded synthetic weights in the reference's checkpoint key layout and synthetic query / support
inputs: what bench.py, the smoke test and the parity tests feed to BOTH the HIP path and the CPU oracle.
Pure data generation: no model arithmetic, no dependency on oracle/.

Key layout (state_dict of sylph MetaOneStageDetector, captured from the reference modules):
  backbone.bottom_up.stem.conv1.{weight,norm.*}, backbone.bottom_up.res{2..5}.{i}.{shortcut,conv1,
  conv2,conv3}.{weight,norm.{weight,bias,running_mean,running_var}}, backbone.fpn_lateral{3,4,5}.*,
  backbone.fpn_output{3,4,5}.*, backbone.top_block.p{6,7}.*,
  proposal_generator.fcos_head.{cls_tower,bbox_tower}.{0,3,6,9}.{weight,bias} (+GN 1,4,7,10),
  ...cls_logits/bbox_pred/ctrness/iou_overlap/scales.{l}.scale  (sylph/modeling/meta_fcos/fcos.py:382-461)
  code_generator.code_generator_head.{support_set_shared_tower.{0,3}(+GN 1,4), post_norm,
  support_set_cls_conv.0, support_set_cls_bias.0, bias_scale.scale, conv_scale.scale, init_norm.{l}}
  (sylph/modeling/code_generator/code_generator.py:277-439)
"""
import math
from typing import Dict, List, Tuple

import torch

STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}  # detectron2 ResNet depths


def _conv(g, cout, cin, k, std=None):
    fan_in = cin * k * k
    std = math.sqrt(2.0 / fan_in) if std is None else std
    return torch.randn(cout, cin, k, k, generator=g) * std


def _bn(g, sd, prefix, c):
    sd[prefix + ".weight"] = torch.rand(c, generator=g) + 0.5
    sd[prefix + ".bias"] = torch.randn(c, generator=g) * 0.1
    sd[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.1
    sd[prefix + ".running_var"] = torch.rand(c, generator=g) + 0.5


def _gn(g, sd, prefix, c):
    sd[prefix + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
    sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=g)


def backbone_state_dict(seed: int = 0, depth: int = 50) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "backbone.bottom_up"
    # inputs are O(100) (0..255 minus mean): scale the stem so activations are O(1)
    sd[f"{p}.stem.conv1.weight"] = _conv(g, 64, 3, 7) / 64.0
    _bn(g, sd, f"{p}.stem.conv1.norm", 64)
    cin = 64
    for si, nb in enumerate(STAGE_BLOCKS[depth]):
        stage = si + 2
        mid, cout = 64 * 2 ** si, 256 * 2 ** si
        for bi in range(nb):
            q = f"{p}.res{stage}.{bi}"
            if bi == 0:
                sd[f"{q}.shortcut.weight"] = _conv(g, cout, cin, 1)
                _bn(g, sd, f"{q}.shortcut.norm", cout)
            sd[f"{q}.conv1.weight"] = _conv(g, mid, cin, 1)
            _bn(g, sd, f"{q}.conv1.norm", mid)
            sd[f"{q}.conv2.weight"] = _conv(g, mid, mid, 3)
            _bn(g, sd, f"{q}.conv2.norm", mid)
            # keep the residual branch small so activations stay O(1) through 16-33 blocks
            sd[f"{q}.conv3.weight"] = _conv(g, cout, mid, 1) * 0.25
            _bn(g, sd, f"{q}.conv3.norm", cout)
            cin = cout
    for stage, c in ((3, 512), (4, 1024), (5, 2048)):
        sd[f"backbone.fpn_lateral{stage}.weight"] = _conv(g, 256, c, 1, std=math.sqrt(1.0 / c))
        sd[f"backbone.fpn_lateral{stage}.bias"] = torch.randn(256, generator=g) * 0.1
        sd[f"backbone.fpn_output{stage}.weight"] = _conv(g, 256, 256, 3, std=math.sqrt(1.0 / 2304))
        sd[f"backbone.fpn_output{stage}.bias"] = torch.randn(256, generator=g) * 0.1
    for n in ("p6", "p7"):
        sd[f"backbone.top_block.{n}.weight"] = _conv(g, 256, 256, 3, std=math.sqrt(1.0 / 2304))
        sd[f"backbone.top_block.{n}.bias"] = torch.randn(256, generator=g) * 0.1
    return sd


def head_state_dict(seed: int = 1, num_classes: int = 60, c: int = 256, num_convs: int = 4,
                    levels: int = 5, num_share_convs: int = 0, norm: str = "GN", num_cls_convs: int = None,
                    num_box_convs: int = None) -> Dict[str, torch.Tensor]:
    """MetaFCOSHead weights under the reference's keys.  norm "GN": a tower is nn.Sequential(conv, GroupNorm, ReLU) x n (indices 3i,
    3i + 1); norm "none": (conv, ReLU) x n (index 2i) -- fcos.py:72-122.  num_share_convs: the shared tower in front of both."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "proposal_generator.fcos_head"
    step = 3 if norm == "GN" else 2
    for t, n in (("cls_tower", num_convs if num_cls_convs is None else num_cls_convs),
                 ("bbox_tower", num_convs if num_box_convs is None else num_box_convs)):  # MODEL.FCOS.NUM_CLS_CONVS / NUM_BOX_CONVS
        for i in range(n):
            sd[f"{p}.{t}.{step * i}.weight"] = _conv(g, c, c, 3, std=math.sqrt(2.0 / (9 * c)))
            sd[f"{p}.{t}.{step * i}.bias"] = torch.randn(c, generator=g) * 0.1
            if norm == "GN":
                _gn(g, sd, f"{p}.{t}.{step * i + 1}", c)
    g2 = torch.Generator().manual_seed(seed * 7 + 3)  # a separate stream: the keys above stay bit-stable when a shared tower is added
    for i in range(num_share_convs):
        sd[f"{p}.share_tower.{step * i}.weight"] = _conv(g2, c, c, 3, std=math.sqrt(2.0 / (9 * c)))
        sd[f"{p}.share_tower.{step * i}.bias"] = torch.randn(c, generator=g2) * 0.1
        if norm == "GN":
            _gn(g2, sd, f"{p}.share_tower.{step * i + 1}", c)
    sd[f"{p}.cls_logits.weight"] = _conv(g, num_classes, c, 1, std=0.01)
    sd[f"{p}.cls_logits.bias"] = torch.full((num_classes,), -math.log(99.0))
    sd[f"{p}.bbox_pred.weight"] = _conv(g, 4, c, 3, std=0.02)
    sd[f"{p}.bbox_pred.bias"] = torch.rand(4, generator=g) * 2.0 + 1.0
    sd[f"{p}.ctrness.weight"] = _conv(g, 1, c, 3, std=0.02)
    sd[f"{p}.ctrness.bias"] = torch.randn(1, generator=g) * 0.1
    sd[f"{p}.iou_overlap.weight"] = _conv(g, 1, c, 3, std=0.02)
    sd[f"{p}.iou_overlap.bias"] = torch.randn(1, generator=g) * 0.1
    for l in range(levels):
        sd[f"{p}.scales.{l}.scale"] = torch.tensor([1.0 + 0.1 * l])
    return sd


def codegen_state_dict(seed: int = 2, c: int = 256, out_c: int = 256, tower_layers: int = 2,
                       levels: int = 5, weight_scale_layers: bool = False, tower_spec=None) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "code_generator.code_generator_head"
    for l in range(levels):
        _gn(g, sd, f"{p}.init_norm.{l}", c)
    # tower_spec: CODE_GENERATOR.TOWER_LAYERS as [[norm, act], ...] (default: tower_layers x ["GN", "ReLU"]); the nn.Sequential index
    # advances by one per module that exists (conv, [norm], [act]) -- code_generator.py:648-688
    spec = tower_spec if tower_spec is not None else [["GN", "ReLU"]] * tower_layers
    idx = 0
    for norm, act in spec:
        sd[f"{p}.support_set_shared_tower.{idx}.weight"] = _conv(g, c, c, 3, std=math.sqrt(2.0 / (9 * c)))
        sd[f"{p}.support_set_shared_tower.{idx}.bias"] = torch.randn(c, generator=g) * 0.1
        idx += 1
        if norm == "GN":
            _gn(g, sd, f"{p}.support_set_shared_tower.{idx}", c)
            idx += 1
        if act in ("ReLU", "Tanh"):
            idx += 1
    _gn(g, sd, f"{p}.post_norm", out_c)
    sd[f"{p}.support_set_cls_conv.0.weight"] = _conv(g, out_c, c, 3, std=math.sqrt(1.0 / (9 * c)))
    sd[f"{p}.support_set_cls_conv.0.bias"] = torch.randn(out_c, generator=g) * 0.1
    sd[f"{p}.support_set_cls_bias.0.weight"] = _conv(g, 1, c, 3, std=math.sqrt(1.0 / (9 * c)))
    sd[f"{p}.support_set_cls_bias.0.bias"] = torch.randn(1, generator=g) * 0.1
    sd[f"{p}.bias_scale.scale"] = torch.tensor([0.9])
    sd[f"{p}.conv_scale.scale"] = torch.tensor([1.3])
    if weight_scale_layers:  # CODE_GENERATOR.WEIGHT_LAYER / SCALE_LAYER heads (own generator: the weights above stay what they were)
        g2 = torch.Generator().manual_seed(seed + 1000)
        for head, bias0 in (("support_set_cls_weight", 0.0), ("support_set_cls_scale", 1.0)):
            sd[f"{p}.{head}.0.weight"] = _conv(g2, 1, c, 3, std=math.sqrt(4.0 / (9 * c)))
            sd[f"{p}.{head}.0.bias"] = torch.randn(1, generator=g2) * 0.1 + bias0
    return sd


def synthetic_state_dict(seed: int = 0, depth: int = 50, num_classes: int = 60) -> Dict[str, torch.Tensor]:
    sd = {"pixel_mean": torch.tensor([103.530, 116.280, 123.675]).view(3, 1, 1),
          "pixel_std": torch.tensor([1.0, 1.0, 1.0]).view(3, 1, 1)}
    sd.update(backbone_state_dict(seed, depth))
    sd.update(head_state_dict(seed + 1, num_classes))
    sd.update(codegen_state_dict(seed + 2))
    return sd


def synthetic_images(n: int, h: int, w: int, seed: int = 1) -> List[torch.Tensor]:
    """uint8-valued fp32 BGR images (3,H,W) in [0,255]."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 256, (3, h, w), generator=g).float() for _ in range(n)]


def synthetic_boxes(n: int, h: int, w: int, seed: int = 2) -> torch.Tensor:
    """One XYXY box per support image: x0,y0 ~ U(0, 0.5*W/H); w,h ~ U(32, 0.5*min)."""
    g = torch.Generator().manual_seed(seed)
    x0 = torch.rand(n, generator=g) * 0.5 * w
    y0 = torch.rand(n, generator=g) * 0.5 * h
    m = 0.5 * min(h, w)
    lo = min(32.0, 0.5 * m)
    bw = lo + torch.rand(n, generator=g) * (m - lo)
    bh = lo + torch.rand(n, generator=g) * (m - lo)
    return torch.stack([x0, y0, x0 + bw, y0 + bh], dim=1)


def synthetic_codes(n: int, c: int = 256, seed: int = 3, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Normalised-looking class codes: unit-L2 rows x scale, bias at the focal prior."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(n, c, 1, 1, generator=g)
    w = w / w.flatten(1).norm(dim=1).view(n, 1, 1, 1) * scale
    return {"cls_conv": w, "cls_bias": torch.full((n,), -math.log(99.0)) + 0.1 * torch.randn(n, generator=g)}


def roi_encoder_state_dict(seed: int = 4, c: int = 256, inter: int = 64, fc_dim: int = 256, head_fc: int = 512,
                           layers: int = 2, tok_convs: int = 2, tok_fcs: int = 2) -> Dict[str, torch.Tensor]:
    """Seeded weights in the key layout of the reference ROIEncoder module
    (sylph/modeling/code_generator/roi_encoder.py:206-281, utils.py:70-141)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    p = "code_generator"

    def lin(name, o, i, std=None):
        sd[f"{name}.weight"] = torch.randn(o, i, generator=g) * (std if std is not None else math.sqrt(1.0 / i))
        sd[f"{name}.bias"] = torch.randn(o, generator=g) * 0.1

    def gn(name, ch):
        sd[f"{name}.weight"] = 1.0 + 0.1 * torch.randn(ch, generator=g)
        sd[f"{name}.bias"] = 0.1 * torch.randn(ch, generator=g)

    sd[f"{p}.box_pooler.conv.0.weight"] = _conv(g, c, c, 3, std=math.sqrt(2.0 / (9 * c)))
    sd[f"{p}.box_pooler.conv.0.bias"] = torch.randn(c, generator=g) * 0.1
    gn(f"{p}.box_pooler.conv.1", c)
    cam = f"{p}.box_pooler.context_attention_module"
    for branch, idx in (("local_att", (0, 1, 3, 4)), ("global_att", (1, 2, 4, 5))):
        sd[f"{cam}.{branch}.{idx[0]}.weight"] = _conv(g, inter, c, 1, std=math.sqrt(1.0 / c))
        sd[f"{cam}.{branch}.{idx[0]}.bias"] = torch.randn(inter, generator=g) * 0.1
        gn(f"{cam}.{branch}.{idx[1]}", inter)
        sd[f"{cam}.{branch}.{idx[2]}.weight"] = _conv(g, c, inter, 1, std=math.sqrt(1.0 / inter))
        sd[f"{cam}.{branch}.{idx[2]}.bias"] = torch.randn(c, generator=g) * 0.1
        gn(f"{cam}.{branch}.{idx[3]}", c)
    for k in range(tok_convs):
        sd[f"{p}.tokenizer.conv{k + 1}.weight"] = _conv(g, c, c, 3, std=math.sqrt(2.0 / (9 * c)))
        gn(f"{p}.tokenizer.conv{k + 1}.norm", c)
    din = c * 49
    for k in range(tok_fcs):
        lin(f"{p}.tokenizer.fc{k + 1}", fc_dim, din, std=math.sqrt(2.0 / din))
        din = fc_dim
    for l in range(layers):
        q = f"{p}.transformer_encoder.layers.{l}"
        sd[f"{q}.self_attn.in_proj_weight"] = torch.randn(3 * fc_dim, fc_dim, generator=g) * math.sqrt(1.0 / fc_dim)
        sd[f"{q}.self_attn.in_proj_bias"] = torch.randn(3 * fc_dim, generator=g) * 0.1
        lin(f"{q}.self_attn.out_proj", fc_dim, fc_dim)
        lin(f"{q}.linear1", 4 * fc_dim, fc_dim, std=math.sqrt(2.0 / fc_dim))
        lin(f"{q}.linear2", fc_dim, 4 * fc_dim)
        gn(f"{q}.norm1", fc_dim)
        gn(f"{q}.norm2", fc_dim)
    lin(f"{p}.weight_head.fc1", head_fc, fc_dim, std=math.sqrt(2.0 / fc_dim))
    lin(f"{p}.weight_head.fc2", c, head_fc)
    lin(f"{p}.bias_head.fc1", head_fc, fc_dim, std=math.sqrt(2.0 / fc_dim))
    lin(f"{p}.bias_head.fc2", 1, head_fc)
    sd["proposal_generator.fcos_head.cond_cls_logits.scales.0.scale"] = torch.tensor([0.8])
    return sd
