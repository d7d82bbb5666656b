"""The HIP path (through the C ABI) against the independent restatements of the third-party half (tests/independent_refs.py: float64
separable-matrix ROIAlignV2 + level assignment, FPN through Hugging Face's Sam2VisionNeck + explicit float64 3x3 sums, numpy
detector_postprocess) -- the same references tests/test_independent_pins.py holds the oracle to.  VERDICT r4 next #5."""
import os

import numpy as np
import pytest
import torch

import independent_refs as IR

pytestmark = pytest.mark.gpu


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


def test_hip_roi_align_matches_separable_float64():
    """sylph_roi_align (roi_align_kernel, level assignment included) on 200 random boxes over random 5-level pyramids of 256 channels:
    every level, all four borders straddled, boxes wholly outside (exact zeros), thin boxes."""
    from sylph_amd import synthetic as W
    eng = _engine()
    eng.load_state_dict(W.codegen_state_dict(seed=2))  # any weights: the context must be finalized
    feats, boxes = IR.random_roi_case(seed=0, S=200, C=256)
    levels = set()
    for lo in range(0, 200, 50):
        f = [x[lo:lo + 50] for x in feats]
        b = boxes[lo:lo + 50]
        want = IR.roi_pool_separable_f64(f, b)
        eng.import_pyramid([torch.from_numpy(x) for x in f], (256, 320))
        got = eng.roi_align(torch.from_numpy(b)).cpu().numpy()
        outside = np.abs(want).reshape(50, -1).max(1) == 0
        assert outside.sum() >= 3
        np.testing.assert_array_equal(got[outside], 0.0)
        np.testing.assert_allclose(got, want, atol=5e-5, rtol=1e-5)
        levels |= {IR.level_of_box(x) for x in b}
    assert levels == {3, 4, 5, 6, 7}


@pytest.mark.parametrize("dtype,rel", [("f32", 1e-3), ("f32s", 1e-3), ("bf16", 3e-2)])
def test_hip_fpn_matches_hf_neck_on_its_own_stage_outputs(dtype, rel):
    """FPN laterals + fused nearest-2x top-down adds + 3x3 output convs + P6 / P7 of the HIP backbone, from the HIP path's OWN res3..res5
    (parity taps), against Sam2VisionNeck + float64 sums: pins the FPN half independently of the ResNet half (that one is pinned
    against transformers.ResNetModel through the oracle)."""
    from sylph_amd import synthetic as W
    sd = W.backbone_state_dict(0, depth=50)
    eng = _engine(dtype)
    eng.load_state_dict(sd)
    imgs = W.synthetic_images(2, 96, 160, seed=21)
    assert eng.preprocess(imgs) == (96, 160)
    eng.backbone()
    res = [eng.export_stage(s).cpu() for s in (3, 4, 5)]
    got = eng.export_pyramid()
    want = IR.fpn_via_hf_neck(res[0], res[1], res[2], sd)
    for l, wnt in enumerate(want):
        g = got[l].cpu().numpy()
        assert g.shape == wnt.shape
        err = np.abs(g - wnt).max()
        assert err <= rel * max(1.0, np.abs(wnt).max()), (l, err, np.abs(wnt).max())


def test_hip_postprocess_matches_numpy_on_random_rescales(golden_dir):
    """detector_postprocess inside sylph_decode_nms under random output sizes: the boxes of a decode at the images' own size, rescaled
    and clipped by the numpy restatement, equal the device's boxes at the other output size (same detections, same order)."""
    from sylph_amd import synthetic as W
    g1 = np.load(os.path.join(golden_dir, "g1_head_decode.npz"))
    eng = _engine(**{"MODEL.FCOS.POST_NMS_TOPK_TEST": 1000, "MODEL.FCOS.NMS_TH": 1.0})  # every candidate comes out: many clipped boxes
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    feats = [torch.from_numpy(g1[f"feat{l}_q8"].astype(np.float32) / 32.0) for l in range(5)]
    eng.import_pyramid(feats, (128, 160), sizes)
    tag = "n20_t50"
    eng.import_head([torch.from_numpy(g1[f"{tag}_logits{l}"]) for l in range(5)], [torch.from_numpy(g1[f"reg{l}"]) for l in range(5)],
                    [torch.from_numpy(g1[f"ctr{l}"]) for l in range(5)], [torch.from_numpy(g1[f"iou{l}"]) for l in range(5)])
    base = eng.decode(sizes, max_out=6000)
    rng = np.random.default_rng(11)
    for _ in range(6):
        outs = [(int(rng.integers(40, 700)), int(rng.integers(40, 900))) for _ in sizes]
        dets = eng.decode(outs, max_out=6000)
        for i, (b0, d) in enumerate(zip(base, dets)):
            wb, keep = IR.postprocess_f64(b0["pred_boxes"].cpu().numpy(), sizes[i], outs[i][0], outs[i][1])
            margin = np.minimum(wb[:, 2] - wb[:, 0], wb[:, 3] - wb[:, 1])
            assert not np.any((margin > 0) & (margin < 1e-3))
            assert keep.sum() > 100
            np.testing.assert_array_equal(d["cand_index"].cpu().numpy(), b0["cand_index"].cpu().numpy()[keep])
            np.testing.assert_allclose(d["pred_boxes"].cpu().numpy(), wb[keep], atol=2e-4, rtol=1e-6)
