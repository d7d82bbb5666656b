"""Third-party half of the oracle (detectron2 ROIPooler / ROIAlignV2, FPN top-down + LastLevelP6P7, detector_postprocess) against
restatements that share no code with oracle/ (tests/independent_refs.py): a float64 separable-matrix ROIAlign on 200 random boxes
over a random 5-level pyramid (borders straddled on all sides, boxes outside, thin boxes), the FPN through Hugging Face's
Sam2VisionNeck + explicit float64 3x3 sums, the postprocess by numpy on random rescales.  VERDICT r4 next #5.  The HIP path runs the
same comparisons in tests/test_independent_pins_gpu.py."""
import numpy as np
import pytest
import torch

import independent_refs as IR
from oracle import backbone as OB
from oracle import decode as OD
from oracle import roi_align as OR
from sylph_amd import synthetic as W


def test_oracle_roi_pooler_matches_separable_float64():
    feats, boxes = IR.random_roi_case(seed=0, S=200)
    want = IR.roi_pool_separable_f64(feats, boxes)
    got = OR.roi_pooler([torch.from_numpy(f) for f in feats], torch.from_numpy(boxes)).numpy()
    lv = np.array([IR.level_of_box(b) for b in boxes])
    assert set(lv.tolist()) == {3, 4, 5, 6, 7}  # every level is exercised
    assert (np.abs(want).reshape(200, -1).max(1) == 0).sum() >= 10  # boxes wholly outside the map: exact zeros expected ...
    np.testing.assert_array_equal(got[np.abs(want).reshape(200, -1).max(1) == 0], 0.0)  # ... and delivered
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)


def test_oracle_level_assignment_matches_formula_on_random_boxes():
    _, boxes = IR.random_roi_case(seed=3, S=400)
    got = OR.assign_boxes_to_levels(torch.from_numpy(boxes), 3, 7).numpy() + 3
    want = np.array([IR.level_of_box(b) for b in boxes])
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("hw", [(64, 96), (96, 160)])
def test_oracle_fpn_matches_hf_neck_and_float64_convs(hw):
    sd = W.backbone_state_dict(0, depth=50)
    g = torch.Generator().manual_seed(5)
    h, w = hw
    res = {"res3": torch.randn(2, 512, h // 8, w // 8, generator=g), "res4": torch.randn(2, 1024, h // 16, w // 16, generator=g),
           "res5": torch.randn(2, 2048, h // 32, w // 32, generator=g)}
    got = OB.fpn(res, sd)
    want = IR.fpn_via_hf_neck(res["res3"], res["res4"], res["res5"], sd)
    for name, wnt in zip(("p3", "p4", "p5", "p6", "p7"), want):
        assert tuple(got[name].shape) == wnt.shape, name
        err = np.abs(got[name].numpy() - wnt).max()
        assert err <= 2e-4 * max(1.0, np.abs(wnt).max()), (name, err)


def test_oracle_postprocess_matches_numpy_on_random_rescales():
    rng = np.random.default_rng(7)
    for _ in range(20):
        ih, iw = int(rng.integers(200, 900)), int(rng.integers(200, 1400))
        oh, ow = int(rng.integers(100, 1200)), int(rng.integers(100, 1600))
        n = 300
        c = np.stack([rng.uniform(-0.1 * iw, 1.1 * iw, n), rng.uniform(-0.1 * ih, 1.1 * ih, n)], 1)
        wh = np.stack([rng.uniform(0, 0.5 * iw, n), rng.uniform(0, 0.5 * ih, n)], 1) * (rng.uniform(size=(n, 1)) > 0.1)
        boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        inst = {"pred_boxes": torch.from_numpy(boxes), "scores": torch.arange(n, dtype=torch.float32)}
        got = OD.detector_postprocess(inst, (ih, iw), oh, ow)
        wb, keep = IR.postprocess_f64(boxes, (ih, iw), oh, ow)
        # an fp32 box whose clipped extent is a rounding error away from empty may fall either way: none in these draws by margin
        margin = np.minimum(wb[:, 2] - wb[:, 0], wb[:, 3] - wb[:, 1])
        assert not np.any((margin > 0) & (margin < 1e-3))
        np.testing.assert_array_equal(got["scores"].numpy(), np.arange(n, dtype=np.float32)[keep])
        np.testing.assert_allclose(got["pred_boxes"].numpy(), wb[keep], atol=2e-4, rtol=1e-6)
