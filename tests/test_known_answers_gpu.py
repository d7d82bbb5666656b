"""G8 known-answer fixtures against the HIP path, through the C ABI (sylph_roi_align, sylph_import_head +
sylph_decode_nms, sylph_conv2d compositions, sylph_backbone_fpn).  The expectations are float64 / hand-derived and
independent of oracle/ (tests/golden/gen_known_answers.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g8(golden_dir):
    return np.load(os.path.join(golden_dir, "g8_known_answers.npz"))


def _engine(dtype="f32", **over):
    from sylph_amd.config import get_default_cfg
    from sylph_amd.engine import Engine
    cfg = get_default_cfg()
    cfg.MODEL.META_LEARN.EPISODIC_LEARNING = True
    cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
    cg.CONV_L2_NORM = True
    cg.TOWER_LAYERS = [["GN", "ReLU"], ["GN", "ReLU"]]
    cg.CLS_LAYER = ["", "", 1]
    cg.BIAS_LAYER = ["", "", 1]
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return Engine(cfg, dtype=dtype)


def _close(got, want, rel):
    err = float(np.abs(got - want).max())
    assert err <= rel * max(1.0, float(np.abs(want).max())), f"max err {err} vs scale {np.abs(want).max()}"


def test_roi_align_and_level_assignment_known_answers(g8):
    from sylph_amd import synthetic as W
    eng = _engine()
    eng.load_state_dict(W.codegen_state_dict(seed=2))  # any weights: the context must be finalized
    S = g8["roi_boxes"].shape[0]
    feats = [torch.from_numpy(g8[f"roi_feat{l}"])[None].repeat(S, 1, 1, 1) for l in range(5)]
    eng.import_pyramid(feats, (256, 256))
    got = eng.roi_align(torch.from_numpy(g8["roi_boxes"])).cpu().numpy()
    # every level holds a different offset (l + 1), so a wrong level assignment is a gross error, not a rounding one
    np.testing.assert_allclose(got, g8["roi_expect"], atol=5e-5, rtol=1e-5)


def test_nms_and_postprocess_known_answers(g8):
    from sylph_amd import synthetic as W
    eng = _engine()
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g8["nms_image_sizes"]]
    outs = [tuple(int(v) for v in s) for s in g8["nms_out_sizes"]]
    shapes = [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    eng.import_pyramid([torch.zeros(2, 256, h, w) for h, w in shapes], (64, 64), sizes)
    eng.import_head([torch.from_numpy(g8[f"nms_logits{l}"]) for l in range(5)],
                    [torch.from_numpy(g8[f"nms_reg{l}"]) for l in range(5)],
                    [torch.from_numpy(g8[f"nms_ctr{l}"]) for l in range(5)],
                    [torch.zeros(2, 1, h, w) for h, w in shapes])
    dets = eng.decode(outs)
    np.testing.assert_array_equal(dets[0]["cand_index"].cpu().numpy(), g8["nms_img0_cand"])
    for i, d in enumerate(dets):
        np.testing.assert_array_equal(d["pred_classes"].cpu().numpy(), g8[f"nms_img{i}_classes"])
        np.testing.assert_array_equal(d["locations"].cpu().numpy(), g8[f"nms_img{i}_locations"])
        np.testing.assert_allclose(d["scores"].cpu().numpy(), g8[f"nms_img{i}_scores"], atol=1e-6)
        np.testing.assert_allclose(d["pred_boxes"].cpu().numpy(), g8[f"nms_img{i}_boxes"], atol=1e-5)


@pytest.fixture(scope="module")
def g8b(golden_dir):
    return np.load(os.path.join(golden_dir, "g8b_known_answers.npz"))


def test_roi_align_boundaries_degenerate_and_integral_bins(g8, g8b):
    """g8b ROI cases through roi_align_kernel: sqrt(area) exactly 112 / 224 / 448 / 896 and just below (the 1e-8 epsilon of
    assign_boxes_to_levels decides the exact ones), zero-area and zero-height boxes, boxes wholly outside the map on either side
    (exact zeros), bins of exactly 2 / 3 feature pixels, a box straddling the right border."""
    from sylph_amd import synthetic as W
    eng = _engine()
    eng.load_state_dict(W.codegen_state_dict(seed=2))
    S = g8b["roi_boxes"].shape[0]
    feats = [torch.from_numpy(g8[f"roi_feat{l}"])[None].repeat(S, 1, 1, 1) for l in range(5)]
    eng.import_pyramid(feats, (256, 256))
    got = eng.roi_align(torch.from_numpy(g8b["roi_boxes"])).cpu().numpy()
    np.testing.assert_allclose(got, g8b["roi_expect"], atol=5e-5, rtol=1e-5)
    for k in (9, 10, 11, 12):
        assert np.all(got[k] == 0.0), k
    # the sample count itself, on a parabola (see gen_known_answers.py)
    q = torch.from_numpy(g8b["quad_feat"])[None]
    feats = [torch.nn.functional.pad(q, (0, 0, 0, 0, 0, 252))] + [torch.zeros(1, 256, 32 >> l, 32 >> l) for l in range(1, 5)]
    eng.import_pyramid(feats, (256, 256))
    got = eng.roi_align(torch.from_numpy(g8b["quad_box"])).cpu().numpy()[0, :4]
    np.testing.assert_allclose(got, g8b["quad_expect"], atol=1e-4, rtol=1e-6)


def test_nms_two_chunks_reversed_ties_and_clipping(g8b):
    """g8b decode cases through decode + nms_kernel: 70 kept boxes of one class (the walk crosses the 64-box chunk boundary), a
    candidate of the second chunk suppressed by a box kept in the FIRST, one suppressed inside the second chunk, one kept for its
    class, IoU == 0.6 (not suppressed) incl. the reversed score order, score ties in both geometric orders (lower ordinal wins),
    clipping at 0 and to an empty box under a 1.25 x 1.5 rescale."""
    from sylph_amd import synthetic as W
    eng = _engine()
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g8b["nms_image_sizes"]]
    outs = [tuple(int(v) for v in s) for s in g8b["nms_out_sizes"]]
    shapes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    eng.import_pyramid([torch.zeros(2, 256, h, w) for h, w in shapes], (128, 128), sizes)
    eng.import_head([torch.from_numpy(g8b[f"nms_logits{l}"]) for l in range(5)],
                    [torch.from_numpy(g8b[f"nms_reg{l}"]) for l in range(5)],
                    [torch.from_numpy(g8b[f"nms_ctr{l}"]) for l in range(5)],
                    [torch.zeros(2, 1, h, w) for h, w in shapes])
    dets = eng.decode(outs)
    np.testing.assert_array_equal(dets[0]["cand_index"].cpu().numpy(), g8b["nms_img0_cand"])
    for i, d in enumerate(dets):
        np.testing.assert_array_equal(d["pred_classes"].cpu().numpy(), g8b[f"nms_img{i}_classes"])
        np.testing.assert_array_equal(d["locations"].cpu().numpy(), g8b[f"nms_img{i}_locations"])
        np.testing.assert_allclose(d["scores"].cpu().numpy(), g8b[f"nms_img{i}_scores"], atol=1e-6)
        np.testing.assert_allclose(d["pred_boxes"].cpu().numpy(), g8b[f"nms_img{i}_boxes"], atol=1e-5)


def _bn(sd, p):
    from oracle import backbone as OB  # scale/shift folding only (host constants for sylph_conv2d)
    return OB.bn_scale_shift(sd, p)


def test_bottleneck_known_answer_f32(g8):
    """res3.0 (projection shortcut, stride 2 on the 1x1s) as four sylph_conv2d calls vs the float64 result."""
    from sylph_amd import synthetic as W
    sd = W.backbone_state_dict(0, depth=50)
    eng = _engine()
    p = "backbone.bottom_up.res3.0"
    x = torch.from_numpy(g8["blk_x"])[None]
    s1, b1 = _bn(sd, p + ".conv1.norm"); s2, b2 = _bn(sd, p + ".conv2.norm")
    s3, b3 = _bn(sd, p + ".conv3.norm"); ss, bs = _bn(sd, p + ".shortcut.norm")
    t = eng.conv2d(x, sd[p + ".conv1.weight"], s1, b1, 2, 0, True)
    t = eng.conv2d(t, sd[p + ".conv2.weight"], s2, b2, 1, 1, True)
    sc = eng.conv2d(x, sd[p + ".shortcut.weight"], ss, bs, 2, 0, False)
    y = eng.conv2d(t, sd[p + ".conv3.weight"], s3, b3, 1, 0, True, residual=sc)
    _close(y[0].cpu().numpy(), g8["blk_y"], 1e-4)


def test_fpn_step_known_answer_f32(g8):
    from sylph_amd import synthetic as W
    sd = W.backbone_state_dict(0, depth=50)
    eng = _engine()
    c4, c5 = torch.from_numpy(g8["fpn_c4"])[None], torch.from_numpy(g8["fpn_c5"])[None]
    ones = lambda n: torch.ones(n)
    prev5 = eng.conv2d(c5, sd["backbone.fpn_lateral5.weight"], ones(256), sd["backbone.fpn_lateral5.bias"])
    up = torch.nn.functional.interpolate(prev5, scale_factor=2.0, mode="nearest")  # data movement only
    inner = eng.conv2d(c4, sd["backbone.fpn_lateral4.weight"], ones(256), sd["backbone.fpn_lateral4.bias"], residual=up)
    _close(inner[0].cpu().numpy(), g8["fpn_inner4"], 1e-4)
    p4 = eng.conv2d(inner, sd["backbone.fpn_output4.weight"], ones(256), sd["backbone.fpn_output4.bias"], 1, 1)
    _close(p4[0].cpu().numpy(), g8["fpn_p4"], 1e-4)


@pytest.mark.parametrize("dtype,rel", [("f32", 1e-3), ("f32s", 1e-3), ("bf16", 6e-2)])
def test_resnet50_fpn_known_answer(g8, dtype, rel):
    """The whole backbone + FPN + P6/P7 (fused top-down upsample, fused projection shortcuts, stem kernel, maxpool) on a
    64 x 96 image vs the float64 result: fp32 mode within the north-star 1e-3; bf16 mode: the bound is the measured
    bf16 rounding accumulation over 53 layers (max error / max |value|), recorded here."""
    from sylph_amd import synthetic as W
    sd = W.backbone_state_dict(0, depth=50)
    eng = _engine(dtype)
    eng.load_state_dict(sd)
    img = W.synthetic_images(1, 64, 96, seed=int(g8["bb_image_seed"]))
    assert eng.preprocess(img) == (64, 96)
    eng.backbone()
    got = eng.export_pyramid()
    for l in range(5):
        _close(got[l][0].cpu().numpy(), g8[f"bb_p{l + 3}"], rel)
