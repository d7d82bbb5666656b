"""Independent restatements of the THIRD-PARTY half of the path (detectron2 / AdelaiDet / torchvision arithmetic that is not under
/root/reference), used to pin oracle/ (CPU tests) and the HIP path (GPU tests) against something their author did not derive case
by case.  TEST INFRASTRUCTURE.  Nothing here imports oracle/ or shares code with it:

  * ROIAlignV2 + FPN level assignment: float64, in a different FORMULATION from oracle/roi_align.py (which loops over samples and
    interpolates each one): the average of bilinear samples over a tensor-product grid is separable, so a pooled map is
    A_y . F . A_x^T with A_y [7 x H] / A_x [7 x W] the per-bin averaged 1-D interpolation weights (rows built from the published
    operator definition: aligned -> -0.5 pixel shift, adaptive grid ceil(roi / 7), samples outside [-1, size] contribute 0,
    coordinates clamped to [0, size - 1]).
  * FPN: the lateral 1x1 convs and the nearest-2x top-down sums by Hugging Face's `transformers` Sam2VisionNeck (an independent
    implementation of exactly that recurrence: prev = lateral(c_i) + upsample(prev)), the 3x3 output convs and LastLevelP6P7 by an
    explicit float64 sliding-window sum (no torch conv).
  * detector_postprocess: numpy float64 re-derivation (scale by out / in per axis, clip to the output size, drop empty boxes).
"""
import math

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ ROIAlignV2
def _axis_weights(lo: float, length: float, size: int, out: int = 7) -> np.ndarray:
    """[out x size] float64: row p = average over the bin's sample points of the 1-D linear-interpolation weights.
    `lo`, `length`: start and extent of the box on this axis in FEATURE coordinates (already scaled and shifted by -0.5)."""
    bin_sz = length / out
    grid = int(math.ceil(bin_sz))  # adaptive sampling ratio (sampling_ratio = 0)
    A = np.zeros((out, size), dtype=np.float64)
    if grid <= 0:
        return A  # no samples: the pooled value is 0 / max(count, 1) = 0
    for p in range(out):
        for i in range(grid):
            c = lo + p * bin_sz + (i + 0.5) * bin_sz / grid
            if c < -1.0 or c > size:
                continue  # the whole sample is zero
            c = max(c, 0.0)
            low = int(c)
            if low >= size - 1:
                A[p, size - 1] += 1.0
            else:
                frac = c - low
                A[p, low] += 1.0 - frac
                A[p, low + 1] += frac
        A[p] /= grid
    return A


def level_of_box(box, min_level=3, max_level=7, canonical_size=224.0, canonical_level=4) -> int:
    """detectron2 assign_boxes_to_levels: floor(4 + log2(sqrt(area) / 224 + 1e-8)) clamped to the pyramid's levels."""
    w, h = float(box[2]) - float(box[0]), float(box[3]) - float(box[1])
    s = math.sqrt(max(w * h, 0.0)) if w * h > 0 else 0.0
    lvl = math.floor(canonical_level + math.log2(s / canonical_size + 1e-8))
    return int(min(max(lvl, min_level), max_level))


def roi_pool_separable_f64(features, boxes, strides=(8, 16, 32, 64, 128), out: int = 7) -> np.ndarray:
    """features[l]: (S, C, h_l, w_l) arrays; boxes (S, 4) xyxy image pixels, box i on image i -> (S, C, out, out) float64."""
    S = boxes.shape[0]
    min_level = int(round(math.log2(strides[0])))
    res = []
    for i in range(S):
        lvl = level_of_box(boxes[i], min_level, min_level + len(strides) - 1) - min_level
        f = np.asarray(features[lvl][i], dtype=np.float64)
        sc = 1.0 / strides[lvl]
        x1, y1, x2, y2 = (float(v) * sc - 0.5 for v in boxes[i])
        Ay = _axis_weights(y1, y2 - y1, f.shape[1], out)
        Ax = _axis_weights(x1, x2 - x1, f.shape[2], out)
        res.append(np.einsum("ph,chw,qw->cpq", Ay, f, Ax))
    return np.stack(res)


# ------------------------------------------------------------------------------------------------ FPN
def conv3x3_f64(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int = 1) -> np.ndarray:
    """x (B, C, H, W), w (O, C, 3, 3), pad 1 -> (B, O, H', W') float64, as nine shifted matrix products."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    B, C, H, Wd = x.shape
    Ho, Wo = (H + 2 - 3) // stride + 1, (Wd + 2 - 3) // stride + 1
    xp = np.zeros((B, C, H + 2, Wd + 2), dtype=np.float64)
    xp[:, :, 1:H + 1, 1:Wd + 1] = x
    out = np.zeros((B, w.shape[0], Ho, Wo), dtype=np.float64)
    for kh in range(3):
        for kw in range(3):
            win = xp[:, :, kh:kh + stride * (Ho - 1) + 1:stride, kw:kw + stride * (Wo - 1) + 1:stride]
            out += np.einsum("oc,bchw->bohw", w[:, :, kh, kw], win)
    return out + np.asarray(b, dtype=np.float64).reshape(1, -1, 1, 1)


def fpn_via_hf_neck(res3, res4, res5, sd, prefix="backbone"):
    """res3..res5: (B, C, h, w) float tensors; sd: detectron2-keyed FPN weights -> [p3, p4, p5, p6, p7] float64 arrays."""
    from transformers.models.sam2.configuration_sam2 import Sam2VisionConfig
    from transformers.models.sam2.modeling_sam2 import Sam2VisionNeck
    chans = [int(res5.shape[1]), int(res4.shape[1]), int(res3.shape[1])]
    cfg = Sam2VisionConfig(backbone_channel_list=chans, fpn_hidden_size=256, fpn_kernel_size=1, fpn_stride=1, fpn_padding=0,
                           fpn_top_down_levels=[0, 1])
    neck = Sam2VisionNeck(cfg).double().eval()
    with torch.no_grad():
        for k, name in enumerate(("fpn_lateral5", "fpn_lateral4", "fpn_lateral3")):  # convs[0] serves the LAST (coarsest) map
            neck.convs[k].weight.copy_(sd[f"{prefix}.{name}.weight"].double())
            neck.convs[k].bias.copy_(sd[f"{prefix}.{name}.bias"].double())
        hidden = [t.double().permute(0, 2, 3, 1) for t in (res3, res4, res5)]  # the neck takes NHWC, finest first
        sums, _ = neck(hidden)  # (coarsest, ..., finest): the top-down sums detectron2 calls prev_features
    inner5, inner4, inner3 = (s.numpy() for s in sums)
    g = lambda n: (sd[f"{prefix}.{n}.weight"].numpy(), sd[f"{prefix}.{n}.bias"].numpy())
    p5 = conv3x3_f64(inner5, *g("fpn_output5"))
    p4 = conv3x3_f64(inner4, *g("fpn_output4"))
    p3 = conv3x3_f64(inner3, *g("fpn_output3"))
    p6 = conv3x3_f64(p5, *g("top_block.p6"), stride=2)
    p7 = conv3x3_f64(np.maximum(p6, 0.0), *g("top_block.p7"), stride=2)
    return [p3, p4, p5, p6, p7]


# ------------------------------------------------------------------------------------------------ detector_postprocess
def postprocess_f64(boxes, image_size, out_h, out_w):
    """boxes (n, 4) xyxy in the network-input frame of an image of size image_size = (h, w) -> (boxes in the out_h x out_w frame,
    keep mask): x scaled by out_w / w, y by out_h / h, clipped to [0, out_w] x [0, out_h], boxes without positive extent dropped."""
    b = np.asarray(boxes, dtype=np.float64).copy()
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b[:, [0, 2]] = np.clip(b[:, [0, 2]] * sx, 0.0, out_w)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]] * sy, 0.0, out_h)
    keep = (b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])
    return b, keep


# ------------------------------------------------------------------------------------------------ shared random cases
def random_roi_case(seed: int = 0, S: int = 200, H: int = 256, W: int = 320, C: int = 8):
    """A random 5-level pyramid (one per box: image i <-> box i) and S boxes: every level populated, boxes straddling all four
    borders, boxes wholly outside, thin boxes, boxes larger than the image."""
    rng = np.random.default_rng(seed)
    feats = [rng.standard_normal((S, C, -(-H // s), -(-W // s))).astype(np.float32) for s in (8, 16, 32, 64, 128)]
    boxes = np.zeros((S, 4), dtype=np.float32)
    for i in range(S):
        kind = i % 10
        size = float(np.exp(rng.uniform(np.log(6.0), np.log(3000.0))))  # sqrt(area) over all five levels
        ar = float(np.exp(rng.uniform(-1.2, 1.2)))
        bw, bh = size * math.sqrt(ar), size / math.sqrt(ar)
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        if kind == 0:
            cx = rng.uniform(-0.2, 0.2) * bw  # straddles the left border
        elif kind == 1:
            cx = W + rng.uniform(-0.2, 0.2) * bw  # right border
        elif kind == 2:
            cy = rng.uniform(-0.2, 0.2) * bh  # top
        elif kind == 3:
            cy = H + rng.uniform(-0.2, 0.2) * bh  # bottom
        elif kind == 4:
            cx, cy = W + bw, H + bh  # wholly outside
        elif kind == 5:
            bh = rng.uniform(0.5, 3.0)  # thin
        boxes[i] = (cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2)
    return feats, boxes
