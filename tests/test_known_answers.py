"""G8 known-answer fixtures (tests/golden/gen_known_answers.py: float64 numpy + hand-derived expectations, independent of
oracle/) against the CPU oracle.  These pin the THIRD-PARTY half of the oracle (detectron2 ROIPooler / ROIAlignV2 /
level assignment, torchvision-style class-aware NMS, detector_postprocess, ResNet bottleneck, FPN, the whole
ResNet-50-FPN) that the reference itself cannot pin.  CPU only; the HIP path is checked against the same
fixtures in tests/test_known_answers_gpu.py."""
import os

import numpy as np
import pytest
import torch

from oracle import backbone as OB
from oracle import decode as OD
from oracle import roi_align as OR
from sylph_amd import synthetic as W


@pytest.fixture(scope="module")
def g8(golden_dir):
    return np.load(os.path.join(golden_dir, "g8_known_answers.npz"))


def test_level_assignment_known_answers(g8):
    lv = OR.assign_boxes_to_levels(torch.from_numpy(g8["roi_boxes"]))
    np.testing.assert_array_equal(lv.numpy(), g8["roi_levels"])  # incl. sqrt(area) == 224 -> level 4, clamps at 3 and 7


def test_roi_align_known_answers(g8):
    S = g8["roi_boxes"].shape[0]
    feats = [torch.from_numpy(g8[f"roi_feat{l}"])[None].repeat(S, 1, 1, 1) for l in range(5)]
    got = OR.roi_pooler(feats, torch.from_numpy(g8["roi_boxes"]))
    np.testing.assert_allclose(got.numpy(), g8["roi_expect"], atol=2e-5, rtol=2e-6)


def test_nms_and_postprocess_known_answers(g8):
    logits = [torch.from_numpy(g8[f"nms_logits{l}"]) for l in range(5)]
    regs = [torch.from_numpy(g8[f"nms_reg{l}"]) for l in range(5)]
    ctrs = [torch.from_numpy(g8[f"nms_ctr{l}"]) for l in range(5)]
    ious = [torch.zeros_like(c) for c in ctrs]
    props = OD.predict_proposals(logits, regs, ctrs, ious)
    for i, p in enumerate(props):
        img, osz = tuple(g8["nms_image_sizes"][i]), tuple(g8["nms_out_sizes"][i])
        if i == 0:  # the NMS keep list itself (IoU exactly 0.6 survives, 0.625 does not; class-aware; tie -> lower index)
            np.testing.assert_array_equal((p["loc_index"] * 3 + p["pred_classes"]).numpy(), g8["nms_img0_cand"])
        r = OD.detector_postprocess(p, img, int(osz[0]), int(osz[1]))
        np.testing.assert_array_equal(r["pred_classes"].numpy(), g8[f"nms_img{i}_classes"])
        np.testing.assert_array_equal(r["locations"].numpy(), g8[f"nms_img{i}_locations"])
        np.testing.assert_allclose(r["scores"].numpy(), g8[f"nms_img{i}_scores"], atol=1e-6)
        np.testing.assert_allclose(r["pred_boxes"].numpy(), g8[f"nms_img{i}_boxes"], atol=1e-5)


# ---- round 4: g8b (level boundaries incl. the 1e-8 epsilon, degenerate / outside boxes, integral adaptive bins, a non-linear map
# that pins the SAMPLE COUNT; NMS over two 64-box chunks, IoU == 0.6 with the score order reversed, ties in both geometric orders,
# clipping at 0 and to empty under a non-square rescale) ------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def g8b(golden_dir):
    return np.load(os.path.join(golden_dir, "g8b_known_answers.npz"))


def test_level_assignment_boundaries(g8b):
    lv = OR.assign_boxes_to_levels(torch.from_numpy(g8b["roi_boxes"]))
    np.testing.assert_array_equal(lv.numpy(), g8b["roi_levels"])


def test_roi_align_degenerate_outside_and_integral_bins(g8, g8b):
    S = g8b["roi_boxes"].shape[0]
    feats = [torch.from_numpy(g8[f"roi_feat{l}"])[None].repeat(S, 1, 1, 1) for l in range(5)]
    got = OR.roi_pooler(feats, torch.from_numpy(g8b["roi_boxes"])).numpy()
    np.testing.assert_allclose(got, g8b["roi_expect"], atol=2e-5, rtol=2e-6)
    for k in (9, 10, 11, 12):  # zero-area, zero-height, wholly outside (both sides): exact zeros, not NaN
        assert np.all(got[k] == 0.0), k


def test_roi_align_sample_count_on_a_parabola(g8b):
    """bin size exactly 2 feature pixels -> ceil(2.0) = 2 samples per bin and direction; on f = x^2 three samples would give another mean."""
    q = torch.from_numpy(g8b["quad_feat"])[None]
    feats = [torch.nn.functional.pad(q, (0, 0, 0, 0, 0, 252))] + [torch.zeros(1, 256, 32 >> l, 32 >> l) for l in range(1, 5)]
    got = OR.roi_pooler(feats, torch.from_numpy(g8b["quad_box"])).numpy()[0, :4]
    np.testing.assert_allclose(got, g8b["quad_expect"], atol=1e-4, rtol=1e-6)


def test_nms_two_chunks_reversed_ties_and_clipping(g8b):
    logits = [torch.from_numpy(g8b[f"nms_logits{l}"]) for l in range(5)]
    regs = [torch.from_numpy(g8b[f"nms_reg{l}"]) for l in range(5)]
    ctrs = [torch.from_numpy(g8b[f"nms_ctr{l}"]) for l in range(5)]
    ious = [torch.zeros_like(c) for c in ctrs]
    props = OD.predict_proposals(logits, regs, ctrs, ious)
    loc_base = np.cumsum([0, 256, 64, 16, 4])
    for i, p in enumerate(props):
        img, osz = tuple(g8b["nms_image_sizes"][i]), tuple(g8b["nms_out_sizes"][i])
        r = OD.detector_postprocess(p, img, int(osz[0]), int(osz[1]))
        np.testing.assert_array_equal(r["pred_classes"].numpy(), g8b[f"nms_img{i}_classes"])
        np.testing.assert_array_equal(r["locations"].numpy(), g8b[f"nms_img{i}_locations"])
        np.testing.assert_allclose(r["scores"].numpy(), g8b[f"nms_img{i}_scores"], atol=1e-6)
        np.testing.assert_allclose(r["pred_boxes"].numpy(), g8b[f"nms_img{i}_boxes"], atol=1e-5)
    p0 = props[0]  # the keep list itself, as global candidate ordinals (level offset + location) * N + class
    ords = (torch.from_numpy(loc_base)[p0["fpn_levels"]] + p0["loc_index"]) * 3 + p0["pred_classes"]
    np.testing.assert_array_equal(ords.numpy(), g8b["nms_img0_cand"])


@pytest.fixture(scope="module")
def bb_sd(g8):
    sd = W.backbone_state_dict(0, depth=50)
    chk = float(sum(v.double().abs().sum() for k, v in sorted(sd.items())))
    assert abs(chk - float(g8["bb_weights_checksum"])) < 1e-3 * chk
    return sd


def _close(got, want, rel):
    err = float(np.abs(got - want).max())
    assert err <= rel * max(1.0, float(np.abs(want).max())), f"max err {err} vs scale {np.abs(want).max()}"


def test_bottleneck_known_answer(g8, bb_sd):
    y = OB.bottleneck(torch.from_numpy(g8["blk_x"])[None], bb_sd, "backbone.bottom_up.res3.0", 2, True)
    _close(y[0].numpy(), g8["blk_y"], 1e-5)


def test_fpn_step_known_answer(g8, bb_sd):
    out = OB.fpn({"res3": torch.zeros(1, 512, 12, 16), "res4": torch.from_numpy(g8["fpn_c4"])[None],
                  "res5": torch.from_numpy(g8["fpn_c5"])[None]}, bb_sd)
    _close(out["p4"][0].numpy(), g8["fpn_p4"], 1e-5)


def test_resnet50_fpn_known_answer(g8, bb_sd):
    img = W.synthetic_images(1, 64, 96, seed=int(g8["bb_image_seed"]))
    x, _ = OB.preprocess(img)
    pyr = OB.backbone_fpn(x, bb_sd, 50)
    for l, p in enumerate(pyr):
        _close(p[0].numpy(), g8[f"bb_p{l + 3}"], 2e-5)
