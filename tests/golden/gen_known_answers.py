"""G8 known-answer fixtures for the THIRD-PARTY primitives of the path (SURVEY.md 8c last row): arithmetic that does not
live under /root/reference (detectron2 ROIPooler / ROIAlignV2 / assign_boxes_to_levels, torchvision nms via adet ml_nms,
adet/detectron2 detector_postprocess, detectron2 ResNet bottleneck + FPN), so the reference cannot pin it.

Everything here is computed WITHOUT the oracle and WITHOUT torch operators: float64 numpy written from the operator
definitions in a different shape than oracle/ (ROIAlign as separable interpolation matrices, convolution as im2col +
matmul, NMS as an O(n^2) suppression table), plus hand-derived expectations (linear feature maps make ROIAlign a closed
form; the NMS / postprocess boxes are constructed so that IoUs and scaled coordinates are exact small rationals).
The oracle (CPU tests) AND the HIP path (-m gpu tests, through the C ABI) are checked against these fixtures.

    python tests/golden/gen_known_answers.py      # writes tests/golden/g8_known_answers.npz and g8b_known_answers.npz
    python tests/golden/gen_known_answers.py --only-b   # only the round-4 additions (g8b: level boundaries, degenerate / outside
                                                        # boxes, integral adaptive bins; two-chunk NMS, reversed IoU == 0.6, ties, clipping)

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


# ===================================================================================== round 4: wider third-party cases (g8b)
ROI_CASES_B = [
    # box, expected level (0-based pyramid index), closed form applies
    ([8.0, 8.0, 120.0, 120.0], 0, True),            # sqrt(area) == 112: 4 + log2(0.5 + 1e-8) = 3.00000003 -> level 3 (the epsilon decides)
    ([8.0, 8.0, 119.9, 119.9], 0, True),            # just below: floor(2.9987) = 2 -> clamped to 3
    ([16.0, 16.0, 239.9, 239.9], 0, True),          # just below the canonical size
    ([16.0, 16.0, 240.0, 240.0], 1, True),          # exactly the canonical size -> level 4
    ([0.0, 0.0, 112.0, 112.0], 0, False),           # starts at the image corner: the first samples sit at -0.5 + ... < 0 and are clamped to 0
    ([-96.0, -96.0, 352.0, 352.0], 2, False),       # sqrt(area) == 448 exactly -> level 5 (samples beyond the map are void)
    ([-96.0, -96.0, 351.9, 351.9], 1, False),       # 447.9 -> level 4
    ([-320.0, -320.0, 576.0, 576.0], 3, False),     # sqrt(area) == 896 exactly -> level 6
    ([-320.0, -320.0, 575.9, 575.9], 2, False),     # 895.9 -> level 5
    ([50.0, 50.0, 50.0, 50.0], 0, False),           # zero-area box: log2(1e-8) -> clamp to level 3; zero bins, zero samples -> all zeros
    ([50.0, 60.0, 90.0, 60.0], 0, False),           # zero height only: no samples either
    ([300.0, 300.0, 400.0, 400.0], 0, False),       # wholly beyond the bottom-right corner of the 256 x 256 map: every sample void
    ([-200.0, -200.0, -100.0, -100.0], 0, False),   # wholly left / above (coordinates < -1): every sample void
    ([16.0, 16.0, 128.0, 128.0], 0, True),          # bin size EXACTLY 2 feature pixels: ceil(2.0) = 2 samples per bin, not 3
    ([16.0, 24.0, 184.0, 192.0], 0, True),          # bin size exactly 3: 3 x 3 samples per bin
    ([8.0, 8.0, 64.0, 120.0], 0, True),             # 1 x 2 feature pixels per bin: non-square adaptive grid (1 x 2 samples)
    ([250.0, 10.0, 262.0, 200.0], 0, False),        # straddles the right border: columns beyond n are void, the row range is inside
]


def gen_roi_align_b(out):
    feats, params = roi_feature_pyramid()
    boxes, levels, expect = [], [], []
    for box, want_l, closed in ROI_CASES_B:
        l = assign_level(box) - 3
        assert l == want_l, (box, l, want_l)
        s = 1.0 / (8 << l)
        got = roi_align_matrix(feats[l], box, s)
        if closed:
            a, b, o = params[l]
            x1, y1, x2, y2 = [v * s - 0.5 for v in box]
            cx = x1 + (np.arange(7) + 0.5) * (x2 - x1) / 7
            cy = y1 + (np.arange(7) + 0.5) * (y2 - y1) / 7
            hand = a[:, None, None] * cx[None, None, :] + b[:, None, None] * cy[None, :, None] + o[:, None, None]
            assert np.abs(hand - got).max() < 1e-10, (box, np.abs(hand - got).max())
        boxes.append(box); levels.append(l); expect.append(got)
    for k in (9, 10, 11, 12):  # degenerate / outside boxes pool to exact zeros
        assert np.abs(expect[k]).max() == 0.0, (boxes[k], np.abs(expect[k]).max())
    # the integral-bin cases really use ceil(bin) samples: one more sample per bin would move the (non-linear) border cases only, so
    # pin the sample count itself on a NON-linear map: f = x^2 on level 3, box [16,16,128,128] -> bins of 2 pixels, 2 samples at
    # bin_start + 0.5 and + 1.5: mean of the interpolated parabola differs between 2 and 3 samples per bin
    C, Hh, Ww = 4, 32, 32
    yy, xx = np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij")
    quad = np.stack([xx.astype(F64) ** 2, yy.astype(F64) ** 2, (xx * yy).astype(F64), (xx + 2.0 * yy).astype(F64)])
    qbox = [16.0, 16.0, 128.0, 128.0]
    qgot = roi_align_matrix(quad, qbox, 1.0 / 8)
    # hand: bin p covers feature x in [1.5 + 2p, 3.5 + 2p); samples at 2 + 2p and 3 + 2p (ON pixel centres): mean of x^2 = ((2+2p)^2 + (3+2p)^2) / 2
    px = np.arange(7)
    hand_x2 = ((2.0 + 2 * px) ** 2 + (3.0 + 2 * px) ** 2) / 2.0
    assert np.abs(qgot[0] - hand_x2[None, :]).max() < 1e-9
    assert np.abs(qgot[1] - hand_x2[:, None]).max() < 1e-9
    out["roi_boxes"] = np.asarray(boxes, np.float32)
    out["roi_levels"] = np.asarray(levels, np.int64)
    out["roi_expect"] = np.asarray(expect).astype(np.float32)
    out["quad_feat"] = quad.astype(np.float32)
    out["quad_box"] = np.asarray([qbox], np.float32)
    out["quad_expect"] = qgot.astype(np.float32)
    print("roi_align (b):", len(boxes), "cases, levels", levels)


def gen_nms_b(out):
    """Image 0 (128 x 128): 70 disjoint class-0 boxes (two 64-box chunks of the sorted pool) + candidates that must be suppressed
    ACROSS the chunk boundary / inside the second chunk / kept because of their class.  Image 1 (64 x 64 in a 128 x 128 pad): IoU
    == 0.6 exactly with the score order reversed, score ties in both geometric orders, clipping at 0, non-square rescale to empty."""
    shapes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    N, B = 3, 2
    logits = [np.full((B, N, h, w), -20.0, F64) for h, w in shapes]
    reg = [np.zeros((B, 4, h, w), F64) for h, w in shapes]
    ctr = [np.full((B, 1, h, w), 20.0, F64) for h, w in shapes]
    loc_base = np.cumsum([0] + [h * w for h, w in shapes])

    def put(b, lvl, i, j, cls, logit, box):
        st = 8 << lvl
        x, y = st * j + st // 2, st * i + st // 2
        l, t, r, bt = (x - box[0]) / st, (y - box[1]) / st, (box[2] - x) / st, (box[3] - y) / st
        assert min(l, t, r, bt) >= 0, (box, x, y)
        assert logits[lvl][b, cls, i, j] == -20.0
        logits[lvl][b, cls, i, j] = logit
        reg[lvl][b, :, i, j] = (l, t, r, bt)
        return {"b": b, "ord": int(loc_base[lvl] + i * shapes[lvl][1] + j) * N + cls, "cls": cls,
                "score": math.sqrt(sigmoid(logit) * sigmoid(20.0)), "box": list(map(float, box)), "xy": (float(x), float(y)), "lvl": lvl}

    def iou(a, b):
        iw = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0); ih = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
        inter = iw * ih
        return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)

    def brute_nms(cands, thr=0.6):
        order = sorted(range(len(cands)), key=lambda k: (-cands[k]["score"], cands[k]["ord"]))
        dead, keep = set(), []
        for a_i, ia in enumerate(order):
            if ia in dead:
                continue
            keep.append(ia)
            for ib in order[a_i + 1:]:
                if ib not in dead and cands[ia]["cls"] == cands[ib]["cls"] and iou(cands[ia]["box"], cands[ib]["box"]) > thr:
                    dead.add(ib)
        return [cands[k] for k in keep], order

    # ---- image 0
    grid = []
    for k in range(70):
        i, j = k // 16, k % 16
        grid.append(put(0, 0, i, j, 0, 3.0 - 0.02 * k, [8 * j, 8 * i, 8 * j + 8, 8 * i + 8]))
    # level-1 locations (16 j + 8, 16 i + 8) are corners of level-0 cells: a box ending at the location overlaps the cell up-left of it
    X = put(0, 1, 0, 0, 0, -1.0, [0.5, 0.5, 8, 8])        # vs grid[0] = [0,0,8,8] (rank 0, chunk 0): IoU 56.25/64 = 0.879 -> suppressed; X ranks 70+ (chunk 1)
    Y = put(0, 1, 2, 1, 0, -1.1, [16.5, 32.5, 24, 40])    # vs grid[66] = cell (i=4, j=2) = [16,32,24,40] (rank 66, chunk 1): suppressed inside chunk 1
    Z = put(0, 1, 0, 1, 1, -1.2, [16.5, 0.5, 24, 8])      # overlaps grid[2] = [16,0,24,8] but is class 1: kept
    V = put(0, 1, 0, 2, 0, -1.3, [35.2, 0, 40, 8])        # vs grid[4] = [32,0,40,8]: IoU = 38.4/64 = 0.6 (binary-rounded just BELOW 0.6 in fp32 and fp64): kept
    assert abs(iou(X["box"], grid[0]["box"]) - 0.87890625) < 1e-12 and iou(Y["box"], grid[66]["box"]) > 0.6
    assert iou(V["box"], grid[4]["box"]) <= 0.6 and np.float32(38.4) / np.float32(64.0) <= np.float32(0.6)
    img0 = grid + [X, Y, Z, V]
    keep0, order0 = brute_nms(img0)
    assert [id(c) for c in keep0] == [id(c) for c in grid + [Z, V]]
    assert order0.index(70) >= 64 and order0.index(71) >= 64 and order0.index(66) >= 64 and order0.index(0) < 64  # the chunk layout the case is about

    # ---- image 1: valid image 64 x 64 (inside the 128 x 128 pad), output 96 (h) x 80 (w): sy = 1.5, sx = 1.25
    A = put(1, 0, 0, 0, 0, 1.0, [0, 0, 40, 40])
    Bh = put(1, 0, 0, 1, 0, 2.0, [0, 0, 40, 24])          # IoU(A, Bh) = 0.6 exactly, Bh scores HIGHER: Bh first, A survives (not > 0.6)
    Ch = put(1, 0, 1, 0, 0, 2.5, [0, 0, 40, 25])          # IoU(A, Ch) = 0.625, Ch higher: A is suppressed by Ch after all; IoU(Bh, Ch) = 0.96: Bh suppressed too
    T1 = put(1, 0, 6, 6, 2, 0.0, [44, 44, 60, 60])        # score tie, lower ordinal, the LARGER box
    T2 = put(1, 0, 6, 7, 2, 0.0, [46, 44, 60, 60])        # IoU = 224/256 = 0.875: suppressed by T1 (lower ordinal wins)
    U1 = put(1, 0, 3, 6, 1, 0.0, [46, 20, 60, 36])        # the same tie with the geometry swapped: lower ordinal has the SMALLER box
    U2 = put(1, 0, 3, 7, 1, 0.0, [44, 20, 60, 36])        # suppressed by U1
    Ng = put(1, 0, 0, 4, 1, 1.5, [-12, -20, 44, 12])      # clipped at 0: -> [0, 0, 44, 12] * (1.25, 1.5) = [0, 0, 55, 18]
    Em = put(1, 0, 7, 0, 2, 0.7, [0, 58, 6, 70])          # beyond the bottom: y0 * 1.5 = 87, y1 -> clip 96; kept: [0, 87, 7.5, 96]
    Eh = put(1, 0, 8, 3, 0, 0.6, [20, 64, 30, 72])        # y0 * 1.5 = 96 = the output height: clipped to an EMPTY box -> dropped
    img1 = [A, Bh, Ch, T1, T2, U1, U2, Ng, Em, Eh]
    keep1_all, _ = brute_nms(img1)
    assert [id(c) for c in keep1_all] == [id(c) for c in [Ch, Ng, Em, Eh, U1, T1]], [img1.index(c) for c in keep1_all]
    sx, sy, ow, oh = 80 / 64.0, 96 / 64.0, 80.0, 96.0
    post1, keep1 = [], []
    for c in keep1_all:
        x0, y0, x1, y1 = c["box"]
        bx = [min(max(x0 * sx, 0.0), ow), min(max(y0 * sy, 0.0), oh), min(max(x1 * sx, 0.0), ow), min(max(y1 * sy, 0.0), oh)]
        if bx[2] - bx[0] > 0 and bx[3] - bx[1] > 0:  # Boxes.nonempty()
            keep1.append(c); post1.append(bx)
    assert [id(c) for c in keep1] == [id(c) for c in [Ch, Ng, Em, U1, T1]]
    assert post1[1] == [0.0, 0.0, 55.0, 18.0] and post1[2] == [0.0, 87.0, 7.5, 96.0]

    for l in range(5):
        out[f"nms_logits{l}"] = logits[l].astype(np.float32)
        out[f"nms_reg{l}"] = reg[l].astype(np.float32)
        out[f"nms_ctr{l}"] = ctr[l].astype(np.float32)
    out["nms_image_sizes"] = np.array([[128, 128], [64, 64]])
    out["nms_out_sizes"] = np.array([[128, 128], [96, 80]])
    for i, (keep, boxes) in enumerate(((keep0, [c["box"] for c in keep0]), (keep1, post1))):
        out[f"nms_img{i}_boxes"] = np.asarray(boxes, np.float32)
        out[f"nms_img{i}_scores"] = np.asarray([c["score"] for c in keep], np.float32)
        out[f"nms_img{i}_classes"] = np.asarray([c["cls"] for c in keep], np.int64)
        out[f"nms_img{i}_locations"] = np.asarray([c["xy"] for c in keep], np.float32)
        out[f"nms_img{i}_cand"] = np.asarray([c["ord"] for c in keep], np.int64)
    print("nms/postprocess (b): image 0 keeps", len(keep0), "of", len(img0), "; image 1 keeps", len(keep1), "of", len(img1))


def write_b():
    out = {}
    gen_roi_align_b(out)
    gen_nms_b(out)
    np.savez_compressed(os.path.join(HERE, "g8b_known_answers.npz"), **out)
    print("wrote g8b_known_answers.npz", os.path.getsize(os.path.join(HERE, "g8b_known_answers.npz")) // 1024, "KiB")


if __name__ == "__main__":
    out = {}
    if "--only-b" not in sys.argv:
        gen_roi_align(out)
        gen_nms_postprocess(out)
        gen_backbone(out)
    if "--only-b" not in sys.argv:
        np.savez_compressed(os.path.join(HERE, "g8_known_answers.npz"), **out)
        print("wrote g8_known_answers.npz", os.path.getsize(os.path.join(HERE, "g8_known_answers.npz")) // 1024, "KiB")
    write_b()
