"""G8 known-answer fixtures for the THIRD-PARTY primitives of the path (SURVEY.md 8c last row): arithmetic that does not
live under /root/reference (detectron2 ROIPooler / ROIAlignV2 / assign_boxes_to_levels, torchvision nms via adet ml_nms,
adet/detectron2 detector_postprocess, detectron2 ResNet bottleneck + FPN), so the reference cannot pin it.

Everything here is computed WITHOUT the oracle and WITHOUT torch operators: float64 numpy written from the operator
definitions in a different shape than oracle/ (ROIAlign as separable interpolation matrices, convolution as im2col +
matmul, NMS as an O(n^2) suppression table), plus hand-derived expectations (linear feature maps make ROIAlign a closed
form; the NMS / postprocess boxes are constructed so that IoUs and scaled coordinates are exact small rationals).
The oracle (CPU tests) AND the HIP path (-m gpu tests, through the C ABI) are checked against these fixtures.

    python tests/golden/gen_known_answers.py      # writes tests/golden/g8_known_answers.npz

Reference call sites: sylph/modeling/code_generator/code_generator.py:341-348,928-930 (ROIPooler),
sylph/modeling/meta_fcos/fcos_outputs.py:15,904-1028 (decode, ml_nms, top-k keep),
sylph/modeling/meta_arch/meta_one_stage_detector.py:75,181,273,288-296 (backbone, detector_postprocess).
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))

F64 = np.float64


# ===================================================================================== ROIAlignV2 + level assignment
def roi_feature_pyramid(H=256, W=256, C=256):
    """Level l (stride 8 << l), channel c: f(y, x) = a_c x + b_c y + o_{l,c}: linear in the pixel index, exactly
    representable in fp32, different per level (so the pooled values reveal the level that was used)."""
    c = np.arange(C)
    a = ((c % 7) - 3) / 8.0
    b = (((c // 7) % 5) - 2) / 8.0
    feats, params = [], []
    for l in range(5):
        s = 8 << l
        h, w = H // s, W // s
        o = (l + 1) + (c % 16) / 16.0
        y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        feats.append((a[:, None, None] * x[None] + b[:, None, None] * y[None] + o[:, None, None]).astype(F64))
        params.append((a, b, o))
    return feats, params


def assign_level(box):
    """detectron2 assign_boxes_to_levels: floor(4 + log2(sqrt(area) / 224 + 1e-8)) clamped to [3, 7]."""
    area = (box[2] - box[0]) * (box[3] - box[1])
    lvl = math.floor(4 + math.log2(math.sqrt(area) / 224.0 + 1e-8))
    return min(max(lvl, 3), 7)


def interp_row(t, n):
    """1-D weights of torchvision roi_align's bilinear_interpolate at coordinate t on n samples (zero row when the
    coordinate is outside [-1, n])."""
    w = np.zeros(n, F64)
    if t < -1.0 or t > n:
        return w
    t = max(t, 0.0)
    lo = int(t)
    if lo >= n - 1:
        lo = hi = n - 1
        t = float(lo)
    else:
        hi = lo + 1
    frac = t - lo
    w[lo] += 1.0 - frac
    w[hi] += frac
    return w


def roi_align_matrix(feat, box, scale, out=7):
    """ROIAlign(aligned=True, sampling_ratio=0) as out = Wy @ feat @ Wx^T / count (the sample value is separable, and a
    sample is void as soon as either coordinate is out of range)."""
    C, H, W = feat.shape
    x1, y1, x2, y2 = [v * scale - 0.5 for v in box]
    bw, bh = (x2 - x1) / out, (y2 - y1) / out
    gw, gh = max(int(math.ceil(bw)), 1), max(int(math.ceil(bh)), 1)  # ceil(roi / out); count = max(gh * gw, 1)
    if math.ceil(bw) <= 0:
        gw = 0
    if math.ceil(bh) <= 0:
        gh = 0
    Wy = np.zeros((out, H), F64)
    Wx = np.zeros((out, W), F64)
    for p in range(out):
        for i in range(gh):
            Wy[p] += interp_row(y1 + p * bh + (i + 0.5) * bh / gh, H)
        for i in range(gw):
            Wx[p] += interp_row(x1 + p * bw + (i + 0.5) * bw / gw, W)
    count = max(gh * gw, 1)
    return np.einsum("py,cyx,qx->cpq", Wy, feat, Wx) / count


ROI_CASES = [
    # box (x0, y0, x1, y1), expected level, closed form applies (all samples strictly inside the map)
    ([16.0, 16.0, 72.0, 72.0], 3, True),           # bins of exactly one feature pixel: samples ON pixel centres
    ([10.0, 10.0, 234.0, 234.0], 4, True),          # sqrt(area) == 224: canonical size -> level 4, 2 x 2 samples per bin
    ([10.0, 10.0, 233.9, 233.9], 3, True),          # just below the canonical size -> level 3
    ([-100.0, -100.0, 348.0, 348.0], 5, False),     # sqrt(area) == 448 -> level 5; samples left/above the map are void
    ([-2000.0, -2000.0, 2000.0, 2000.0], 7, False),  # level clamp at 7 (2 x 2 map), most samples void
    ([100.0, 100.0, 101.0, 101.0], 3, True),        # level clamp at 3, bins far smaller than a pixel
    ([33.3, 47.7, 190.1, 120.9], 3, True),          # fractional box, 3 x 2 samples per bin
    ([200.0, 180.0, 256.0, 256.0], 3, False),       # touches the bottom-right border: coordinates clamped to n - 1
    ([240.0, 240.0, 300.0, 300.0], 3, False),       # partly beyond the border: void samples on the far side
]


def gen_roi_align(out):
    feats, params = roi_feature_pyramid()
    for l, f in enumerate(feats):
        out[f"roi_feat{l}"] = f.astype(np.float32)
    boxes, levels, expect = [], [], []
    for box, want_level, closed in ROI_CASES:
        lvl = assign_level(box)
        assert lvl == want_level, (box, lvl, want_level)
        l = lvl - 3
        s = 1.0 / (8 << l)
        got = roi_align_matrix(feats[l], box, s)
        if closed:  # bilinear interpolation of a linear map is exact: bin average = the map at the bin centre
            a, b, o = params[l]
            x1, y1, x2, y2 = [v * s - 0.5 for v in box]
            cx = x1 + (np.arange(7) + 0.5) * (x2 - x1) / 7
            cy = y1 + (np.arange(7) + 0.5) * (y2 - y1) / 7
            hand = a[:, None, None] * cx[None, None, :] + b[:, None, None] * cy[None, :, None] + o[:, None, None]
            assert np.abs(hand - got).max() < 1e-10, (box, np.abs(hand - got).max())
        boxes.append(box)
        levels.append(l)
        expect.append(got)
    out["roi_boxes"] = np.asarray(boxes, np.float32)
    out["roi_levels"] = np.asarray(levels, np.int64)
    out["roi_expect"] = np.asarray(expect).astype(np.float32)
    print("roi_align:", len(boxes), "cases, levels", levels)


# ===================================================================================== decode + NMS + postprocess
def sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


def gen_nms_postprocess(out):
    """Two 64 x 64 padded images, 3 classes.  Head outputs are -20 logits (never candidates) except the listed entries;
    ctrness +20 everywhere (sigmoid = 1 - 2e-9).  Boxes: location (8j+4, 8i+4) on level 0 -+ reg * 8."""
    H = W = 64
    shapes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    N, B = 3, 2
    logits = [np.full((B, N, h, w), -20.0, F64) for h, w in shapes]
    reg = [np.zeros((B, 4, h, w), F64) for h, w in shapes]
    ctr = [np.full((B, 1, h, w), 20.0, F64) for h, w in shapes]

    def put(b, i, j, cls, logit, box):  # level 0 only: exact reg for the wanted box
        x, y = 8 * j + 4, 8 * i + 4
        l, t, r, bt = (x - box[0]) / 8.0, (y - box[1]) / 8.0, (box[2] - x) / 8.0, (box[3] - y) / 8.0
        assert min(l, t, r, bt) >= 0
        logits[0][b, cls, i, j] = logit
        reg[0][b, :, i, j] = (l, t, r, bt)
        return {"b": b, "loc": i * 8 + j, "cls": cls, "score": math.sqrt(sigmoid(logit) * sigmoid(20.0)), "box": list(map(float, box)),
                "xy": (float(x), float(y))}

    # ---- image 0: NMS semantics (image size = padded size, output size = image size)
    A = put(0, 0, 0, 0, 2.0, [0, 0, 40, 40])
    Bx = put(0, 0, 1, 0, 1.0, [0, 0, 40, 24])     # IoU(A, B) = 960 / 1600 = 0.6 exactly: NOT suppressed (needs > 0.6)
    C = put(0, 1, 0, 0, 0.5, [0, 0, 40, 25])      # IoU(A, C) = 1000 / 1600 = 0.625: suppressed by A
    D = put(0, 1, 1, 1, 1.5, [0, 0, 40, 40])      # same box as A, other class: kept (class-aware)
    E1 = put(0, 4, 4, 2, 0.0, [20, 20, 52, 52])   # score tie with E2, identical boxes: the lower candidate index survives
    E2 = put(0, 4, 5, 2, 0.0, [20, 20, 52, 52])
    Fb = put(0, 7, 7, 1, -2.9, [56, 56, 64, 64])  # sigmoid(-2.9) = 0.0522 > 0.05: a candidate
    put(0, 7, 0, 1, -3.0, [0, 56, 8, 64])         # sigmoid(-3.0) = 0.0474 < 0.05: never a candidate
    img0 = [A, Bx, C, D, E1, E2, Fb]
    # hand-derived keep list, in the order ml_nms returns it (descending score; ties by candidate index):
    keep0 = [A, D, Bx, E1, Fb]
    # ---- image 1: detector_postprocess semantics: image 48 x 56 inside the 64 x 64 pad, output 96 x 84 (sy = 2, sx = 1.5)
    Fp = put(1, 2, 2, 0, 2.0, [10, 10, 30, 30])   # -> [15, 20, 45, 60]
    G = put(1, 5, 6, 1, 1.0, [44, 36, 68, 60])    # -> [66, 72, 102, 120] -> clipped to [66, 72, 84, 96]
    Hh = put(1, 1, 7, 2, 0.5, [58, 8, 62, 16])    # -> [87, 16, 93, 32] -> clipped to [84, 16, 84, 32]: empty, dropped
    keep1 = [Fp, G]
    post1 = [[15.0, 20.0, 45.0, 60.0], [66.0, 72.0, 84.0, 96.0]]

    # independent cross-check of the hand-derived keep list: O(n^2) suppression table in float64
    def brute_nms(cands, thr=0.6):
        order = sorted(range(len(cands)), key=lambda k: (-cands[k]["score"], cands[k]["loc"] * N + cands[k]["cls"]))
        dead, keep = set(), []
        for a_i, ia in enumerate(order):
            if ia in dead:
                continue
            keep.append(ia)
            for ib in order[a_i + 1:]:
                ca, cb = cands[ia], cands[ib]
                if ib in dead or ca["cls"] != cb["cls"]:
                    continue
                xa, xb = ca["box"], cb["box"]
                iw = max(min(xa[2], xb[2]) - max(xa[0], xb[0]), 0.0)
                ih = max(min(xa[3], xb[3]) - max(xa[1], xb[1]), 0.0)
                inter = iw * ih
                union = (xa[2] - xa[0]) * (xa[3] - xa[1]) + (xb[2] - xb[0]) * (xb[3] - xb[1]) - inter
                if inter / union > thr:
                    dead.add(ib)
        return [cands[k] for k in keep]

    assert [id(c) for c in brute_nms(img0)] == [id(c) for c in keep0]
    assert [id(c) for c in brute_nms([Fp, G, Hh])] == [id(c) for c in [Fp, G, Hh]]

    for l in range(5):
        out[f"nms_logits{l}"] = logits[l].astype(np.float32)
        out[f"nms_reg{l}"] = reg[l].astype(np.float32)
        out[f"nms_ctr{l}"] = ctr[l].astype(np.float32)
    out["nms_image_sizes"] = np.array([[64, 64], [48, 56]])
    out["nms_out_sizes"] = np.array([[64, 64], [96, 84]])
    for i, (keep, boxes) in enumerate(((keep0, [c["box"] for c in keep0]), (keep1, post1))):
        out[f"nms_img{i}_boxes"] = np.asarray(boxes, np.float32)
        out[f"nms_img{i}_scores"] = np.asarray([c["score"] for c in keep], np.float32)
        out[f"nms_img{i}_classes"] = np.asarray([c["cls"] for c in keep], np.int64)
        out[f"nms_img{i}_locations"] = np.asarray([c["xy"] for c in keep], np.float32)
        out[f"nms_img{i}_cand"] = np.asarray([c["loc"] * N + c["cls"] for c in keep], np.int64)
    print("nms/postprocess: image 0 keeps", len(keep0), "of", len(img0), "; image 1 keeps", len(keep1), "of 3")


# ===================================================================================== ResNet bottleneck / FPN / backbone
def conv2d(x, w, stride=1, pad=0):
    """x (C,H,W), w (O,C,k,k) float64 -> (O,Ho,Wo): im2col + one matmul."""
    C, H, W = x.shape
    O, _, k, _ = w.shape
    xp = np.zeros((C, H + 2 * pad, W + 2 * pad), F64)
    xp[:, pad:pad + H, pad:pad + W] = x
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    cols = np.empty((C, k, k, Ho, Wo), F64)
    for i in range(k):
        for j in range(k):
            cols[:, i, j] = xp[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride]
    return (w.reshape(O, -1) @ cols.reshape(C * k * k, Ho * Wo)).reshape(O, Ho, Wo)


def frozen_bn(y, sd, p):
    g, b, m, v = [sd[f"{p}.{k}"].double().numpy() for k in ("weight", "bias", "running_mean", "running_var")]
    s = g / np.sqrt(v + 1e-5)
    return y * s[:, None, None] + (b - m * s)[:, None, None]


def wnp(sd, k):
    return sd[k].double().numpy()


def bottleneck(x, sd, p, stride, shortcut):
    o = np.maximum(frozen_bn(conv2d(x, wnp(sd, p + ".conv1.weight"), stride), sd, p + ".conv1.norm"), 0)  # STRIDE_IN_1X1
    o = np.maximum(frozen_bn(conv2d(o, wnp(sd, p + ".conv2.weight"), 1, 1), sd, p + ".conv2.norm"), 0)
    o = frozen_bn(conv2d(o, wnp(sd, p + ".conv3.weight")), sd, p + ".conv3.norm")
    s = frozen_bn(conv2d(x, wnp(sd, p + ".shortcut.weight"), stride), sd, p + ".shortcut.norm") if shortcut else x
    return np.maximum(o + s, 0)


def maxpool3s2(x):
    C, H, W = x.shape
    xp = np.full((C, H + 2, W + 2), -np.inf, F64)
    xp[:, 1:-1, 1:-1] = x
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    o = np.full((C, Ho, Wo), -np.inf, F64)
    for i in range(3):
        for j in range(3):
            o = np.maximum(o, xp[:, i:i + 2 * Ho:2, j:j + 2 * Wo:2])
    return o


def backbone_fpn(img, sd, blocks=(3, 4, 6, 3)):
    mean = np.array([103.530, 116.280, 123.675], F64)[:, None, None]
    x = img - mean
    p = "backbone.bottom_up"
    x = np.maximum(frozen_bn(conv2d(x, wnp(sd, p + ".stem.conv1.weight"), 2, 3), sd, p + ".stem.conv1.norm"), 0)
    x = maxpool3s2(x)
    res = {}
    for si, nb in enumerate(blocks):
        for bi in range(nb):
            x = bottleneck(x, sd, f"{p}.res{si + 2}.{bi}", 2 if (bi == 0 and si > 0) else 1, bi == 0)
        res[si + 2] = x

    def cb(x, name, stride=1, pad=0):
        return conv2d(x, wnp(sd, f"backbone.{name}.weight"), stride, pad) + wnp(sd, f"backbone.{name}.bias")[:, None, None]

    prev = cb(res[5], "fpn_lateral5")
    outs = {5: cb(prev, "fpn_output5", 1, 1)}
    for st in (4, 3):
        up = prev.repeat(2, axis=1).repeat(2, axis=2)  # nearest 2x
        prev = cb(res[st], f"fpn_lateral{st}") + up
        outs[st] = cb(prev, f"fpn_output{st}", 1, 1)
    outs[6] = cb(outs[5], "top_block.p6", 2, 1)
    outs[7] = cb(np.maximum(outs[6], 0), "top_block.p7", 2, 1)
    return [outs[k] for k in (3, 4, 5, 6, 7)], res


def gen_backbone(out):
    import torch
    from sylph_amd import synthetic as W
    sd = W.backbone_state_dict(0, depth=50)
    out["bb_weights_checksum"] = float(sum(v.double().abs().sum() for k, v in sorted(sd.items())))
    img = W.synthetic_images(1, 64, 96, seed=77)[0].double().numpy()
    out["bb_image_seed"] = np.array(77)
    pyr, res = backbone_fpn(img, sd)
    for l, p in enumerate(pyr):
        out[f"bb_p{l + 3}"] = p.astype(np.float32)
    # one bottleneck with projection shortcut (res3.0: 256 -> 128 -> 512, stride 2) on a small seeded input
    g = torch.Generator().manual_seed(5)
    xb = torch.randn(256, 10, 12, generator=g).double().numpy()
    out["blk_x"] = xb.astype(np.float32)
    out["blk_y"] = bottleneck(xb, sd, "backbone.bottom_up.res3.0", 2, True).astype(np.float32)
    # one FPN top-down step: p4 = output4(lateral4(res4) + up2(lateral5(res5)))
    c4 = torch.randn(1024, 6, 8, generator=g).double().numpy()
    c5 = torch.randn(2048, 3, 4, generator=g).double().numpy()

    def cb(x, name, stride=1, pad=0):
        return conv2d(x, wnp(sd, f"backbone.{name}.weight"), stride, pad) + wnp(sd, f"backbone.{name}.bias")[:, None, None]

    prev5 = cb(c5, "fpn_lateral5")
    inner4 = cb(c4, "fpn_lateral4") + prev5.repeat(2, axis=1).repeat(2, axis=2)
    out["fpn_c4"], out["fpn_c5"] = c4.astype(np.float32), c5.astype(np.float32)
    out["fpn_inner4"] = inner4.astype(np.float32)
    out["fpn_p4"] = cb(inner4, "fpn_output4", 1, 1).astype(np.float32)
    print("backbone: p3..p7", [p.shape for p in pyr], "max |p3|", float(np.abs(pyr[0]).max()))


if __name__ == "__main__":
    out = {}
    gen_roi_align(out)
    gen_nms_postprocess(out)
    gen_backbone(out)
    np.savez_compressed(os.path.join(HERE, "g8_known_answers.npz"), **out)
    print("wrote g8_known_answers.npz", os.path.getsize(os.path.join(HERE, "g8_known_answers.npz")) // 1024, "KiB")
