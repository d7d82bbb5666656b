"""Oracle: FPN level assignment + ROIAlignV2 (aligned=True, adaptive sampling).  TEST INFRASTRUCTURE.

Third-party arithmetic (detectron2 ``modeling/poolers.py`` ROIPooler / assign_boxes_to_levels,
``layers/roi_align.py`` -> torchvision ``ops.roi_align``; not under /root/reference, parity
unpinned by the reference) restated from the published operator definition.  Reference call sites:
  * sylph/modeling/code_generator/code_generator.py:341-348  ROIPooler(output_size=7,
    scales=[1/s], sampling_ratio=0, pooler_type="ROIAlignV2")
  * sylph/modeling/code_generator/code_generator.py:928-930  box_pooler(features, box_ls)
  * sylph/modeling/code_generator/utils.py:27-47            select_a_mask (one box per image)
"""
import math
from typing import List, Sequence

import numpy as np
import torch

CANONICAL_BOX_SIZE = 224
CANONICAL_LEVEL = 4


def assign_boxes_to_levels(boxes: torch.Tensor, min_level: int = 3, max_level: int = 7,
                           canonical_box_size: int = CANONICAL_BOX_SIZE,
                           canonical_level: int = CANONICAL_LEVEL) -> torch.Tensor:
    """floor(canonical_level + log2(sqrt(area)/canonical_box_size + 1e-8)) clamped, minus min_level."""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lvl = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    lvl = torch.clamp(lvl, min=min_level, max=max_level)
    return lvl.to(torch.int64) - min_level


def _bilinear(feat: np.ndarray, y: float, x: float) -> np.ndarray:
    """feat (C,H,W) -> (C,).  torchvision roi_align bilinear_interpolate."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros((C,), dtype=np.float32)
    y = max(y, 0.0)
    x = max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = np.float32(y - y_low), np.float32(x - x_low)
    hy, hx = np.float32(1.0) - ly, np.float32(1.0) - lx
    w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
    return (w1 * feat[:, y_low, x_low] + w2 * feat[:, y_low, x_high]
            + w3 * feat[:, y_high, x_low] + w4 * feat[:, y_high, x_high]).astype(np.float32)


def roi_align_single(feat: torch.Tensor, box: Sequence[float], spatial_scale: float, out_size: int = 7,
                     sampling_ratio: int = 0, aligned: bool = True) -> torch.Tensor:
    """feat (C,H,W); box xyxy in image pixels -> (C,out,out).  fp32 arithmetic throughout."""
    f = feat.numpy().astype(np.float32)
    C = f.shape[0]
    off = np.float32(0.5 if aligned else 0.0)
    sc = np.float32(spatial_scale)
    x1 = np.float32(box[0]) * sc - off
    y1 = np.float32(box[1]) * sc - off
    x2 = np.float32(box[2]) * sc - off
    y2 = np.float32(box[3]) * sc - off
    rw, rh = np.float32(x2 - x1), np.float32(y2 - y1)
    if not aligned:
        rw, rh = max(rw, np.float32(1.0)), max(rh, np.float32(1.0))
    bw, bh = np.float32(rw / np.float32(out_size)), np.float32(rh / np.float32(out_size))
    gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(bh)))
    gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(bw)))
    count = np.float32(max(gh * gw, 1))
    out = np.zeros((C, out_size, out_size), dtype=np.float32)
    for ph in range(out_size):
        for pw in range(out_size):
            acc = np.zeros((C,), dtype=np.float32)
            for iy in range(gh):
                yy = y1 + np.float32(ph) * bh + (np.float32(iy) + np.float32(0.5)) * bh / np.float32(gh)
                for ix in range(gw):
                    xx = x1 + np.float32(pw) * bw + (np.float32(ix) + np.float32(0.5)) * bw / np.float32(gw)
                    acc += _bilinear(f, float(yy), float(xx))
            out[:, ph, pw] = acc / count
    return torch.from_numpy(out)


def roi_pooler(features: List[torch.Tensor], boxes: torch.Tensor, strides=(8, 16, 32, 64, 128),
               out_size: int = 7) -> torch.Tensor:
    """detectron2 ROIPooler with ONE box per image (image i <-> boxes[i]); features[l] is
    (S,C,h_l,w_l).  Returns (S,C,out,out) in box order."""
    S = boxes.shape[0]
    min_level = int(round(-math.log2(1.0 / strides[0])))
    max_level = int(round(-math.log2(1.0 / strides[-1])))
    if len(features) == 1:
        lvls = torch.zeros(S, dtype=torch.int64)
    else:
        lvls = assign_boxes_to_levels(boxes, min_level, max_level)
    out = []
    for i in range(S):
        l = int(lvls[i])
        out.append(roi_align_single(features[l][i], boxes[i].tolist(), 1.0 / strides[l], out_size))
    return torch.stack(out, dim=0)
