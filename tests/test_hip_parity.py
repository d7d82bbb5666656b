"""GPU parity tests: the HIP path (through the C ABI) against torch fp32 references of single ops,
against the golden vectors generated from the reference, and against the CPU oracle end to end."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _engine(dtype, cfg=None, **kw):
    from sylph_amd.engine import Engine
    if dtype == "f32":  # tests/test_split_mode_gpu.py re-runs the fp32 checks of this file in the split-bf16 parity mode ("f32s")
        dtype = os.environ.get("SYLPH_TEST_F32_MODE", "f32")
    return Engine(cfg, dtype=dtype, **kw)


def _f32_tol(exact, split):
    """Tolerance of an op-level fp32 check: `exact` for the fp32-MFMA mode, `split` when the file runs in the split-bf16 mode (each
    operand carries 16 mantissa bits: 2^-17 relative per product)."""
    return split if os.environ.get("SYLPH_TEST_F32_MODE", "f32") == "f32s" else exact


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _cfg(lvis=False, **over):
    from sylph_amd.config import get_default_cfg
    cfg = get_default_cfg()
    cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
    cfg.MODEL.META_LEARN.EPISODIC_LEARNING = True
    cg.CONV_L2_NORM = True
    cg.TOWER_LAYERS = [["GN", "ReLU"], ["GN", "ReLU"]]
    cg.CLS_LAYER = ["", "", 1]
    cg.BIAS_LAYER = ["", "", 1]
    if lvis:
        cg.BIAS_L2_NORM = True
        cfg.MODEL.FCOS.POST_NMS_TOPK_TEST = 300
    for k, v in over.items():
        cfg.merge_from_list([k, v])
    return cfg


CONV_CASES = [
    # C, Cout, k, stride, pad, H, W, B, relu, residual
    (64, 64, 1, 1, 0, 20, 24, 2, True, False),
    (64, 256, 1, 1, 0, 17, 13, 1, False, True),
    (256, 128, 1, 2, 0, 18, 22, 2, True, False),
    (128, 128, 3, 1, 1, 15, 19, 2, True, False),
    (256, 256, 3, 1, 1, 13, 21, 1, False, False),
    (256, 256, 3, 2, 1, 13, 21, 2, False, False),
    (256, 6, 3, 1, 1, 9, 11, 2, False, False),
    (256, 20, 1, 1, 0, 16, 20, 2, False, False),
    (512, 1024, 1, 1, 0, 8, 10, 1, True, True),
    (256, 256, 3, 1, 1, 7, 7, 5, True, False),
    # conv3 + same-geometry residual of the identity blocks (K 128 / 256 / 512): conv_igemm here (few tiles), conv_spw_kernel under
    # SYLPH_CONV_SPW=2 (tests/test_conv_variants_gpu.py): ragged rows (221 / 189 / 77 per image: partial last tiles), ReLU on and off
    (128, 512, 1, 1, 0, 17, 13, 2, True, True),
    (256, 1024, 1, 1, 0, 9, 21, 3, False, True),
    (512, 2048, 1, 1, 0, 7, 11, 2, True, True),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_matches_torch(dtype, case):
    C, Cout, k, stride, pad, H, W, B, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Cout, C, k, k, generator=g) / (C * k * k) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    eng = _engine(dtype)
    y = eng.conv2d(x, w, scale, shift, stride, pad, relu, res).cpu()
    if dtype == "bf16":
        x, w = _bf16_round(x), _bf16_round(w)
        res = _bf16_round(res) if res is not None else None
    ref = F.conv2d(x, w, None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    tol = _f32_tol(2e-5, 6e-5) if dtype == "f32" else 2e-2
    err = (y - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f"max err {err}"


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 7, 7), (1, 33, 41), (3, 100, 21)])
def test_group_norm_matches_torch(dtype, shape):
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, 256, H, W, generator=g) * 2.0 + 0.7
    gamma = 1 + 0.1 * torch.randn(256, generator=g)
    beta = 0.1 * torch.randn(256, generator=g)
    eng = _engine(dtype)
    y = eng.group_norm(x, gamma, beta, relu=True).cpu()
    xr = _bf16_round(x) if dtype == "bf16" else x
    ref = F.relu(F.group_norm(xr, 32, gamma, beta, eps=1e-5))
    tol = 1e-5 if dtype == "f32" else 2e-2
    assert (y - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


# --------------------------------------------------------------------------------- goldens (reference)
def _feats(g, prefix="feat"):
    return [torch.from_numpy(g[f"{prefix}{l}_q8"].astype(np.float32) / 32.0) for l in range(5)]


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, "g1_head_decode.npz"))


@pytest.fixture(scope="module")
def g3(golden_dir):
    return np.load(os.path.join(golden_dir, "g3_codegen.npz"))


@pytest.fixture(scope="module")
def head_engine():
    from sylph_amd import synthetic as W
    eng = _engine("f32", _cfg())
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    return eng


@pytest.mark.parametrize("tag", ["n1_t50", "n5_t50", "n20_t50"])
def test_head_matches_reference_golden(g1, head_engine, tag):
    eng = head_engine
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.head(torch.from_numpy(g1[f"{tag}_cls_conv"]), torch.from_numpy(g1[f"{tag}_cls_bias"]))
    lo, rg, ct, io = eng.export_head()
    for l in range(5):
        np.testing.assert_allclose(lo[l].cpu().numpy(), g1[f"{tag}_logits{l}"], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(rg[l].cpu().numpy(), g1[f"reg{l}"], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(ct[l].cpu().numpy(), g1[f"ctr{l}"], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(io[l].cpu().numpy(), g1[f"iou{l}"], atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tag,share,norm,owd", [("share1", 1, "GN", False), ("nonorm", 0, "none", False), ("share2_nonorm", 2, "none", False),
                                                ("owd", 0, "GN", True)])
def test_head_variants_match_reference_golden(g1, golden_dir, tag, share, norm, owd, dtype):
    """Reference branches the five target yamls leave off (missing #3), against goldens generated from the reference
    (g1c): MODEL.FCOS.NUM_SHARE_CONVS = 1 / 2 (shared tower in front of the cls / bbox towers, fcos.py:397,626), MODEL.FCOS.NORM "none"
    (conv + ReLU towers, fcos.py:72-122,399) and MODEL.PROPOSAL_GENERATOR.OWD (one all-ones class, fcos_outputs.py:913-916).  fp32: head
    outputs <= 1e-3 and identical (level, location, class) triples; bf16: head outputs to bf16 tolerance."""
    from oracle.decode import detector_postprocess
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g1c_head_variants.npz"))
    cfg = _cfg(**{"MODEL.FCOS.NUM_SHARE_CONVS": share, "MODEL.FCOS.NORM": norm, "MODEL.PROPOSAL_GENERATOR.OWD": owd})
    eng = _engine(dtype, cfg)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60, num_share_convs=share, norm=norm))
    sizes = [tuple(int(v) for v in s) for s in g["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.head(torch.from_numpy(g["cls_conv"]), torch.from_numpy(g["cls_bias"]))
    lo, rg, ct, io = eng.export_head()
    tol = 1e-3 if dtype == "f32" else 6e-2
    for l in range(5):
        if not owd:  # OWD: the conv output is irrelevant (the reference overwrites the probabilities); ours is the constant 40
            ref = g[f"{tag}_logits{l}"]
            assert np.abs(lo[l].cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()), f"logits level {l}"
        for name, got in (("reg", rg), ("ctr", ct)):
            ref = g[f"{tag}_{name}{l}"]
            assert np.abs(got[l].cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()), f"{name} level {l}"
    if dtype != "f32":
        return
    dets = eng.decode()
    for i, d in enumerate(dets):
        pre = f"{tag}_img{i}"
        ref = {k: torch.from_numpy(g[f"{pre}_{k}"]) for k in ("pred_boxes", "scores", "pred_classes", "fpn_levels", "locations")}
        ref = detector_postprocess(ref, sizes[i], sizes[i][0], sizes[i][1])
        assert d["scores"].numel() == ref["scores"].numel() > 0
        np.testing.assert_array_equal(d["pred_classes"].cpu().numpy(), ref["pred_classes"].numpy())
        np.testing.assert_array_equal(d["fpn_levels"].cpu().numpy(), ref["fpn_levels"].numpy())
        np.testing.assert_array_equal(d["locations"].cpu().numpy(), ref["locations"].numpy())
        np.testing.assert_allclose(d["scores"].cpu().numpy(), ref["scores"].numpy(), atol=1e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tag,nc,nb,share,norm", [("c2b3", 2, 3, 0, "GN"), ("c1b4_share1", 1, 4, 1, "GN"), ("c3b1_nonorm", 3, 1, 0, "none"),
                                                   ("c0b2", 0, 2, 0, "GN")])
def test_unequal_tower_depths_match_reference_golden(g1, golden_dir, tag, nc, nb, share, norm, dtype):
    """MODEL.FCOS.NUM_CLS_CONVS != NUM_BOX_CONVS (the HIP path stacks the two towers into one launch per layer only for equal depths:
    unequal depths take the un-paired plan), with / without the shared tower and GroupNorm, a depth-0 cls tower -- against goldens
    generated from the reference (g1e).  fp32: head outputs <= 1e-3 and the reference's (level, location, class) set; bf16: head
    outputs to bf16 tolerance.  VERDICT r4 next #1 (re-audit of the round-4 branch additions)."""
    from oracle.decode import detector_postprocess
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g1e_tower_depths.npz"))
    cfg = _cfg(**{"MODEL.FCOS.NUM_CLS_CONVS": nc, "MODEL.FCOS.NUM_BOX_CONVS": nb, "MODEL.FCOS.NUM_SHARE_CONVS": share, "MODEL.FCOS.NORM": norm})
    eng = _engine(dtype, cfg)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60, num_share_convs=share, norm=norm, num_cls_convs=nc, num_box_convs=nb))
    sizes = [tuple(int(v) for v in s) for s in g["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.head(torch.from_numpy(g["cls_conv"]), torch.from_numpy(g["cls_bias"]))
    lo, rg, ct, io = eng.export_head()
    tol = 1e-3 if dtype == "f32" else 6e-2
    for l in range(5):
        for name, got in (("logits", lo), ("reg", rg), ("ctr", ct), ("iou", io)):
            ref = g[f"{tag}_{name}{l}"]
            assert np.abs(got[l].cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()), f"{name} level {l}"
    if dtype != "f32":
        return
    dets = eng.decode()
    key = lambda x: np.lexsort((x["pred_classes"].cpu().numpy(), x["locations"].cpu().numpy()[:, 0], x["locations"].cpu().numpy()[:, 1],
                                x["fpn_levels"].cpu().numpy()))
    for i, d in enumerate(dets):
        pre = f"{tag}_img{i}"
        ref = {k: torch.from_numpy(g[f"{pre}_{k}"]) for k in ("pred_boxes", "scores", "pred_classes", "fpn_levels", "locations")}
        ref = detector_postprocess(ref, sizes[i], sizes[i][0], sizes[i][1])
        # (the HIP head's outputs differ from the reference's by ~1e-6: near-tied scores may swap places in the sorted output and one
        #  candidate at the post-NMS cut may differ, so the sets are compared with a tolerance of one detection)
        og, orf = key(d), key(ref)
        gk = set(zip(d["fpn_levels"].cpu().numpy()[og].tolist(), map(tuple, d["locations"].cpu().numpy()[og].tolist()), d["pred_classes"].cpu().numpy()[og].tolist()))
        rk = set(zip(ref["fpn_levels"].numpy()[orf].tolist(), map(tuple, ref["locations"].numpy()[orf].tolist()), ref["pred_classes"].numpy()[orf].tolist()))
        assert len(gk ^ rk) <= 2 and abs(len(gk) - len(rk)) <= 1 and len(rk) > 0, (len(gk), len(rk), len(gk ^ rk))


OWD_CASES = [("ctr", ["ctrness"], False, 0.05, 0.6, 100), ("iou", ["iou"], False, 0.05, 0.6, 100), ("ctriou", ["ctrness", "iou"], False, 0.05, 0.6, 100),
             ("ctr_twc", ["ctrness"], True, 0.05, 0.6, 100), ("ctr_t20", ["ctrness"], False, 0.02, 0.6, 100),
             ("ctr_all", ["ctrness"], False, 0.05, 1.0, 1000), ("ctriou_all", ["ctrness", "iou"], False, 0.05, 1.0, 1000),
             ("ctr_top300", ["ctrness"], False, 0.05, 0.6, 300)]


@pytest.mark.parametrize("via", ["head", "import"])
@pytest.mark.parametrize("tag,bq,twc,thr,nms,post", OWD_CASES)
def test_owd_decode_matches_reference_golden(g1, golden_dir, tag, bq, twc, thr, nms, post, via):
    """MODEL.PROPOSAL_GENERATOR.OWD against the adversarial reference golden g1d (quality logits straddling logit(0.05) on every level;
    VERDICT r4 #1): the reference multiplies the all-ones class by the box quality BEFORE the threshold (fcos_outputs.py:937,951).
    `via = "head"`: the HIP head (fp32) on g1's pyramid + decode; `via = "import"`: the reference's own head outputs through
    sylph_import_head + decode (exact).  The `*_all` cases (NMS_TH 1, no post-NMS cut) make the whole candidate set the output: the
    per-level counts of the HIP decode must equal the reference's per-level candidate counts."""
    from oracle.decode import detector_postprocess
    from test_oracle_golden import owd_head_state_dict
    g = np.load(os.path.join(golden_dir, "g1d_owd_decode.npz"))
    cfg = _cfg(**{"MODEL.PROPOSAL_GENERATOR.OWD": True, "MODEL.FCOS.BOX_QUALITY": bq, "MODEL.FCOS.THRESH_WITH_CTR": twc,
                  "MODEL.FCOS.INFERENCE_TH_TEST": thr, "MODEL.FCOS.NMS_TH": nms, "MODEL.FCOS.POST_NMS_TOPK_TEST": post})
    eng = _engine("f32", cfg)
    eng.load_state_dict(owd_head_state_dict())
    sizes = [tuple(int(v) for v in s) for s in g["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    if via == "head":
        eng.head(torch.from_numpy(g["cls_conv"]), torch.from_numpy(g["cls_bias"]))
        lo, rg, ct, io = eng.export_head()
        for l in range(5):
            for name, got in (("reg", rg), ("ctr", ct), ("iou", io)):
                ref = g[f"{name}{l}"]
                assert np.abs(got[l].cpu().numpy() - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max()), f"{name} level {l}"
    else:
        # the all-ones class of the reference (after its `ones_like`): logit 40 -> sigmoid == 1.0f exactly
        ones = [torch.full((2, 1) + g[f"ctr{l}"].shape[-2:], 40.0) for l in range(5)]
        eng.import_head(ones, *[[torch.from_numpy(g[f"{k}{l}"]) for l in range(5)] for k in ("reg", "ctr", "iou")])
    dets = eng.decode(max_out=1200)
    for i, d in enumerate(dets):
        pre = f"{tag}_img{i}"
        ref = {k: torch.from_numpy(g[f"{pre}_{k}"]) for k in ("pred_boxes", "scores", "pred_classes", "fpn_levels", "locations")}
        n_ref_raw = ref["scores"].numel()
        ref = detector_postprocess(ref, sizes[i], sizes[i][0], sizes[i][1])
        assert d["scores"].numel() == ref["scores"].numel() > 0
        if tag.endswith("_all") and ref["scores"].numel() == n_ref_raw:  # nothing suppressed, cut or emptied by the clip: candidates per level
            got_counts = np.bincount(d["fpn_levels"].cpu().numpy(), minlength=5)
            np.testing.assert_array_equal(got_counts, g[f"{tag}_level_counts"][i])
        order_g = order_r = slice(None)
        if via == "head":
            # the HIP head's quality logits differ from the reference's by ~1e-6: near-tied scores may swap places in the score-sorted
            # output, so the SET of (level, location, class) detections is compared (the imported-head variant compares the order too)
            key = lambda x: np.lexsort((x["pred_classes"].cpu().numpy(), x["locations"].cpu().numpy()[:, 0], x["locations"].cpu().numpy()[:, 1],
                                        x["fpn_levels"].cpu().numpy()))
            order_g, order_r = key(d), key(ref)
        pick = lambda x, k, o: x[k].cpu().numpy()[o]
        np.testing.assert_array_equal(pick(d, "pred_classes", order_g), pick(ref, "pred_classes", order_r))
        np.testing.assert_array_equal(pick(d, "fpn_levels", order_g), pick(ref, "fpn_levels", order_r))
        np.testing.assert_array_equal(pick(d, "locations", order_g), pick(ref, "locations", order_r))
        np.testing.assert_allclose(pick(d, "scores", order_g), pick(ref, "scores", order_r), atol=1e-3)
        np.testing.assert_allclose(pick(d, "pred_boxes", order_g), pick(ref, "pred_boxes", order_r), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("tag,thr", [("n1_t50", 0.05), ("n5_t50", 0.05), ("n20_t50", 0.05), ("n20_t11", 0.011)])
def test_decode_matches_reference_golden(g1, tag, thr):
    """boxes/scores within 1e-3 and identical kept (level, location, class) triples."""
    from sylph_amd import synthetic as W
    eng = _engine("f32", _cfg(**{"MODEL.FCOS.INFERENCE_TH_TEST": thr}))
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.head(torch.from_numpy(g1[f"{tag}_cls_conv"]), torch.from_numpy(g1[f"{tag}_cls_bias"]))
    dets = eng.decode()
    from oracle.decode import detector_postprocess
    for i, d in enumerate(dets):
        pre = f"{tag}_img{i}"
        # the golden is predict_proposals' output; the C ABI also applies detector_postprocess
        # (meta_one_stage_detector.py:288-296), restated by the oracle
        ref = {k: torch.from_numpy(g1[f"{pre}_{k}"]) for k in
               ("pred_boxes", "scores", "pred_classes", "fpn_levels", "locations")}
        ref = detector_postprocess(ref, sizes[i], sizes[i][0], sizes[i][1])
        assert d["scores"].numel() == ref["scores"].numel()
        np.testing.assert_array_equal(d["pred_classes"].cpu().numpy(), ref["pred_classes"].numpy())
        np.testing.assert_array_equal(d["fpn_levels"].cpu().numpy(), ref["fpn_levels"].numpy())
        np.testing.assert_array_equal(d["locations"].cpu().numpy(), ref["locations"].numpy())
        np.testing.assert_allclose(d["scores"].cpu().numpy(), ref["scores"].numpy(), atol=1e-3)
        np.testing.assert_allclose(d["pred_boxes"].cpu().numpy(), ref["pred_boxes"].numpy(), atol=1e-3, rtol=1e-4)


DECODE_VARIANTS = [("iou", ["iou"], False), ("ctriou", ["ctrness", "iou"], False), ("ctr_twc", ["ctrness"], True),
                   ("iou_twc", ["iou"], True), ("ctriou_twc", ["ctrness", "iou"], True)]


@pytest.mark.parametrize("tag,bq,twc", DECODE_VARIANTS)
def test_decode_variants_match_reference_golden(g1, golden_dir, tag, bq, twc):
    """quality_mode 1 / 2 (MODEL.FCOS.BOX_QUALITY ["iou"], ["ctrness","iou"]) and THRESH_WITH_CTR of the device decode
    (fcos_outputs.py:938-959): the reference's own head outputs go in through sylph_import_head, the kept
    (level, location, class) triples must be the reference's, scores / boxes within 1e-3."""
    from oracle.decode import detector_postprocess
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g1b_decode_variants.npz"))
    eng = _engine("f32", _cfg(**{"MODEL.FCOS.BOX_QUALITY": bq, "MODEL.FCOS.THRESH_WITH_CTR": twc}))
    assert int(eng.sc.quality_mode) == {"iou": 1, "ctriou": 2, "ctr": 0}[tag.split("_")[0]] and int(eng.sc.thresh_with_ctr) == int(twc)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.import_head([torch.from_numpy(g1[f"n5_t50_logits{l}"]) for l in range(5)], [torch.from_numpy(g1[f"reg{l}"]) for l in range(5)],
                    [torch.from_numpy(g1[f"ctr{l}"]) for l in range(5)], [torch.from_numpy(g1[f"iou{l}"]) for l in range(5)])
    dets = eng.decode()
    for i, d in enumerate(dets):
        pre = f"{tag}_img{i}"
        ref = {k: torch.from_numpy(g[f"{pre}_{k}"]) for k in ("pred_boxes", "scores", "pred_classes", "fpn_levels", "locations")}
        ref = detector_postprocess(ref, sizes[i], sizes[i][0], sizes[i][1])
        assert d["scores"].numel() == ref["scores"].numel() > 0
        np.testing.assert_array_equal(d["pred_classes"].cpu().numpy(), ref["pred_classes"].numpy())
        np.testing.assert_array_equal(d["fpn_levels"].cpu().numpy(), ref["fpn_levels"].numpy())
        np.testing.assert_array_equal(d["locations"].cpu().numpy(), ref["locations"].numpy())
        np.testing.assert_allclose(d["scores"].cpu().numpy(), ref["scores"].numpy(), atol=1e-3)
        np.testing.assert_allclose(d["pred_boxes"].cpu().numpy(), ref["pred_boxes"].numpy(), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("lvis", [False, True])
@pytest.mark.parametrize("S", [1, 2, 5])
def test_codegen_matches_reference_golden(g3, lvis, S):
    from sylph_amd import synthetic as W
    eng = _engine("f32", _cfg(lvis))
    eng.load_state_dict(W.codegen_state_dict(seed=2))
    eng.import_pyramid(_feats(g3, f"s{S}_feat"), (192, 256))
    code = eng.codegen(torch.from_numpy(g3[f"s{S}_boxes"])).cpu().numpy()
    tag = f"{'lvis' if lvis else 'coco'}_s{S}"
    np.testing.assert_allclose(code[:256], g3[f"{tag}_cls_conv"].reshape(-1), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(code[256], g3[f"{tag}_cls_bias"].reshape(-1)[0], atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("tagc", ["coco", "lvis"])
def test_normalize_matches_reference_golden(g3, tagc):
    from sylph_amd import synthetic as W
    eng = _engine("f32", _cfg(tagc == "lvis"))
    eng.load_state_dict(W.codegen_state_dict(seed=2))
    codes = torch.stack([torch.cat([torch.from_numpy(g3[f"{tagc}_s{S}_cls_conv"]).reshape(-1),
                                    torch.from_numpy(g3[f"{tagc}_s{S}_cls_bias"]).reshape(-1)]) for S in (1, 2, 5)])
    out = eng.normalize_codes(codes.cuda().contiguous()).cpu().numpy()
    for i in range(3):
        np.testing.assert_allclose(out[i, :256], g3[f"{tagc}_norm{i}_cls_conv"].reshape(-1), atol=1e-5, rtol=1e-4)
        np.testing.assert_allclose(out[i, 256], g3[f"{tagc}_norm{i}_cls_bias"].reshape(-1)[0], atol=1e-5, rtol=1e-4)


# --------------------------------------------------------------------------------- oracle, end to end
@pytest.fixture(scope="module")
def full_sd():
    from sylph_amd import synthetic as W
    return W.synthetic_state_dict(0, depth=50)


def test_backbone_fpn_matches_oracle_f32(full_sd):
    from oracle import backbone as OB
    from sylph_amd import synthetic as W
    imgs = W.synthetic_images(2, 120, 150, seed=5)
    imgs[1] = imgs[1][:, :97, :131].contiguous()
    eng = _engine("f32", _cfg())
    eng.load_state_dict(full_sd)
    H, Wd = eng.preprocess(imgs)
    assert (H, Wd) == (128, 160)
    eng.backbone()
    got = eng.export_pyramid()
    x, _ = OB.preprocess(imgs)
    ref = OB.backbone_fpn(x, full_sd, 50)
    for l in range(5):
        r = ref[l]
        err = (got[l].cpu() - r).abs().max().item()
        assert err <= 1e-3 * max(1.0, r.abs().max().item()), f"level {l}: {err}"


def test_backbone_fpn_bf16_close_to_oracle(full_sd):
    from oracle import backbone as OB
    from sylph_amd import synthetic as W
    imgs = W.synthetic_images(1, 128, 160, seed=6)
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(full_sd)
    eng.preprocess(imgs)
    eng.backbone()
    got = eng.export_pyramid()
    x, _ = OB.preprocess(imgs)
    ref = OB.backbone_fpn(x, full_sd, 50)
    for l in range(5):
        a, b = got[l].cpu().flatten(), ref[l].flatten()
        cos = F.cosine_similarity(a, b, dim=0).item()
        assert cos > 0.995, f"level {l}: cosine {cos}"


def _episode_codes(eng, sd, n_cls, shots, h, w, oracle_too=True):
    from oracle import episode as E, codegen as CG
    from sylph_amd import synthetic as W
    codes_gpu, codes_ref = [], []
    for c in range(n_cls):
        sup = W.synthetic_images(shots, h, w, seed=50 + c)
        boxes = W.synthetic_boxes(shots, h, w, seed=70 + c)
        eng.preprocess(sup)
        eng.backbone()
        codes_gpu.append(eng.codegen(boxes))
        if oracle_too:
            codes_ref.append(E.forward_class_code(sup, boxes, sd))
    g = eng.normalize_codes(torch.stack(codes_gpu).contiguous())
    ref = None
    if oracle_too:
        recs = [{"support_set_target": torch.tensor(i), "class_name": str(i), "class_code": c}
                for i, c in enumerate(codes_ref)]
        ref = E.format_class_codes_shared(CG.forward_normalize_code(recs, sd))
    return g, ref


def test_full_episode_matches_oracle_f32(full_sd):
    """C1-shaped episode (5-way 1-shot, 2 queries) at small size: codes, boxes, scores <= 1e-3 and the
    same kept (level, location, class) candidates."""
    from oracle import episode as E
    from sylph_amd import synthetic as W
    eng = _engine("f32", _cfg())
    eng.load_state_dict(full_sd)
    g, ref = _episode_codes(eng, full_sd, 5, 1, 128, 160)
    np.testing.assert_allclose(g[:, :256].cpu().numpy(), ref["cls_conv"].reshape(5, 256).numpy(), atol=1e-3)
    np.testing.assert_allclose(g[:, 256].cpu().numpy(), ref["cls_bias"].numpy(), atol=1e-3)
    # make the synthetic detector fire: scale the codes, same scale on both sides
    scale = 3.0
    q = W.synthetic_images(2, 128, 160, seed=9)
    codes_ref = {"cls_conv": ref["cls_conv"] * scale, "cls_bias": ref["cls_bias"]}
    want = E.forward_instances(q, codes_ref, full_sd)
    eng.preprocess(q)
    eng.backbone()
    eng.head(ref["cls_conv"] * scale, ref["cls_bias"])  # identical codes -> isolates the query path
    got = eng.decode()
    assert sum(w["scores"].numel() for w in want) > 20
    for wv, gv in zip(want, got):
        assert gv["scores"].numel() == wv["scores"].numel()
        perm = _same_candidates(gv["cand_index"].cpu().numpy(), _cand_ordinals(wv, 128, 160, 5), wv["scores"].numpy())
        np.testing.assert_allclose(gv["scores"].cpu().numpy(), wv["scores"].numpy()[perm], atol=1e-3)
        _assert_boxes(gv["pred_boxes"].cpu().numpy(), wv["pred_boxes"].numpy()[perm], wv["fpn_levels"].numpy()[perm])
        np.testing.assert_array_equal(gv["pred_classes"].cpu().numpy(), wv["pred_classes"].numpy()[perm])


def _same_candidates(got_ord, want_ord, want_scores):
    """The kept (level, location, class) candidates, in order.  Exact-fp32 mode: identical arrays.  Split-bf16 mode (2^-17 relative per
    product instead of 2^-24): the same SET, and two candidates may trade places only if the oracle's own scores for them are within 1e-4
    of each other (a tie at the mode's resolution; the sort is by score).  -> for each HIP detection its position in the oracle's list."""
    if np.array_equal(got_ord, want_ord):
        return np.arange(want_ord.size)
    assert os.environ.get("SYLPH_TEST_F32_MODE", "f32") == "f32s", (got_ord, want_ord)
    assert got_ord.size == want_ord.size and np.array_equal(np.sort(got_ord), np.sort(want_ord)), "different candidate sets"
    pos = {int(o): k for k, o in enumerate(want_ord.tolist())}
    perm = np.array([pos[int(o)] for o in got_ord.tolist()])
    moved = np.nonzero(perm != np.arange(perm.size))[0]
    gap = np.abs(want_scores[perm[moved]] - want_scores[moved]).max()
    print(f"split mode: {moved.size} candidates trade places, oracle scores within {gap:.2e}")
    assert gap <= 1e-4, gap
    return perm


def _keys_of_ordinals(ords, H, W, N):
    """HIP candidate ordinals -> {(level, location index, class)}."""
    base, bases = 0, []
    h, w = H // 8, W // 8
    for _ in range(5):
        bases.append(base)
        base += h * w
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = set()
    for o in np.asarray(ords).tolist():
        cls, lc = int(o) % N, int(o) // N
        lvl = max(l for l in range(5) if bases[l] <= lc)
        out.add((lvl, lc - bases[lvl], cls))
    return out


def _prove_residue(heads_a, keys_a, heads_b, keys_b, i, what, eps_val, eps_iou, rel=False, **decode_kw):
    """Every detection that only ONE of two pipelines reports for image i must sit within tolerance of a decision boundary of the
    pipeline that lacks it (oracle.decode.explain_absence: the 0.05 threshold, the per-level top-1000 cut, the post-NMS top-k cut, an
    IoU at 0.6 / a score-order flip with its suppressor, or a suppressor that is itself such a case).  -> histogram of the reasons."""
    from collections import Counter
    from oracle import decode as OD
    reasons, bad = Counter(), []
    for missing_in, heads, keys, other in (("b", heads_b, keys_a - keys_b, keys_a), ("a", heads_a, keys_b - keys_a, keys_b)):
        for k in sorted(keys):
            reason, margin, ok = OD.explain_absence(heads, i, k, other, eps_val, eps_iou, rel, **decode_kw)
            reasons[reason] += 1
            if not ok or reason == "present":
                bad.append((missing_in, k, reason, margin))
    print(f"{what}: {len(keys_a ^ keys_b)} differing detections, all marginal: {dict(reasons)}" if not bad else f"{what}: UNEXPLAINED {bad}")
    assert not bad, f"{what}: detections that differ without sitting on a decision boundary: {bad}"
    return reasons


def _cand_ordinals(inst, H, W, N):
    """(level, location, class) ordinal used by the HIP path, from oracle fields."""
    base, bases = 0, []
    h, w = H // 8, W // 8
    for _ in range(5):
        bases.append(base)
        base += h * w
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    lv = inst["fpn_levels"].numpy()
    return (np.asarray(bases)[lv] + inst["loc_index"].numpy()) * N + inst["pred_classes"].numpy()


def test_full_size_properties_bf16(full_sd):
    """BASELINE config C2 shape (800x1333 -> 800x1344, 5-way): size-independent properties."""
    from sylph_amd import synthetic as W
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(full_sd)
    q = W.synthetic_images(2, 800, 1333, seed=3)
    codes = W.synthetic_codes(5, seed=4, scale=3.0)
    assert eng.preprocess(q) == (800, 1344)
    eng.backbone()
    pyr = eng.export_pyramid()
    assert [tuple(p.shape[2:]) for p in pyr] == [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    assert all(torch.isfinite(p).all() for p in pyr)
    eng.head(codes["cls_conv"], codes["cls_bias"])
    d1 = eng.decode()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    d2 = eng.decode()
    for a, b in zip(d1, d2):  # determinism
        assert torch.equal(a["cand_index"], b["cand_index"]) and torch.equal(a["scores"], b["scores"])
    for d in d1:
        s = d["scores"]
        assert s.numel() > 0 and torch.isfinite(s).all() and (s[:-1] >= s[1:]).all()
        bx = d["pred_boxes"]
        assert (bx[:, 0] >= 0).all() and (bx[:, 2] <= 1333).all() and (bx[:, 1] >= 0).all() and (bx[:, 3] <= 800).all()
        assert ((bx[:, 2] - bx[:, 0]) > 0).all() and ((bx[:, 3] - bx[:, 1]) > 0).all()
        # NMS invariant: no same-class pair above the IoU threshold (checked on unclipped geometry is
        # stricter than needed; clipped boxes only shrink intersections proportionally) -> use 0.6 + slack
        bxc, cl = bx.cpu(), d["pred_classes"].cpu()
        area = (bxc[:, 2] - bxc[:, 0]) * (bxc[:, 3] - bxc[:, 1])
        lt = torch.max(bxc[:, None, :2], bxc[None, :, :2])
        rb = torch.min(bxc[:, None, 2:], bxc[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        iou = inter / (area[:, None] + area[None, :] - inter)
        same = (cl[:, None] == cl[None, :]) & ~torch.eye(len(cl), dtype=torch.bool)
        assert (iou[same] <= 0.75).all()


@pytest.mark.parametrize("nb", [120, 144])
def test_full_size_batch120_far_end_of_the_tensors_bf16(full_sd, nb):
    """The default bench batch (120 images of 800x1333: activations of up to 4.13e9 bytes, just below 4 GiB, 1.9 x the signed 32-bit
    range) and 144 images (4.95e9 bytes: past the 32-bit range -- since round 6 bottleneck64[p] and conv_pw address their res2-sized
    operands with a 64-bit base per image + 32-bit offsets inside it; before, those layers fell back to slower kernels above 124
    images).  Four distinct images repeated nb / 4 times: every copy of an image must
    come out BIT FOR BIT the same wherever it sits in the batch (round 6: conv_hpipe pairs patches per image and keys its K-walk
    rotation on the pair's place inside the image, as conv_pw does since round 5: a served image's detections do not depend on its
    neighbours), and equal its 4-image run (other tile shapes at B = 4) up to bf16 rounding."""
    from sylph_amd import synthetic as W
    base = W.synthetic_images(4, 800, 1333, seed=11)
    codes = W.synthetic_codes(5, seed=4, scale=3.0)
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(full_sd)
    assert eng.preprocess([base[i % 4] for i in range(nb)]) == (800, 1344)
    eng.backbone()
    pyr = eng.export_pyramid()
    for lvl, p in enumerate(pyr):
        assert torch.isfinite(p).all()
        for k in range(1, nb // 4):
            assert torch.equal(p[0:4], p[4 * k:4 * k + 4]), f"level {lvl}: copy {k} differs from copy 0 by {(p[0:4] - p[4 * k:4 * k + 4]).abs().max().item()}"
    eng.head(codes["cls_conv"], codes["cls_bias"])
    det = eng.decode()
    assert len(det) == nb
    for i in range(4, nb):
        a, b = det[i], det[i % 4]
        assert a["scores"].numel() == b["scores"].numel() > 0
        assert torch.equal(a["cand_index"], b["cand_index"]) and torch.equal(a["scores"], b["scores"]) and torch.equal(a["pred_boxes"], b["pred_boxes"]), \
            f"image {i}: detections differ from its copy {i % 4}"
    small = [p[0:4].clone() for p in pyr]
    del pyr
    eng4 = _engine("bf16", _cfg())
    eng4.load_state_dict(full_sd)
    eng4.preprocess(base)
    eng4.backbone()
    for lvl, (a, b) in enumerate(zip(small, eng4.export_pyramid())):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 4e-2 * scale, f"level {lvl}: batch-{nb} pyramid vs batch-4 pyramid"


# --------------------------------------------------------------------------------- ROIEncoder (C5)
def _roienc_cfg():
    cfg = _cfg(True)
    cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
    cg.NAME = "ROIEncoder"
    cg.TOKENIZER.NUM_CONV, cg.TOKENIZER.CONV_DIM, cg.TOKENIZER.NORM = 2, 256, "GN"
    cg.TOKENIZER.NUM_FC, cg.TOKENIZER.FC_DIM = 2, 256
    cg.TRANSFORMER_ENCODER.LAYERS, cg.TRANSFORMER_ENCODER.HEADS = 2, 8
    cg.HEAD.NUM_FC, cg.HEAD.FC_DIM, cg.HEAD.OUTPUT_DIM = 2, 512, 256
    return cfg


@pytest.mark.parametrize("S", [2, 5])
def test_roi_encoder_matches_reference_golden(golden_dir, S):
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g7_roi_encoder.npz"))
    eng = _engine("f32", _roienc_cfg())
    eng.load_state_dict(W.roi_encoder_state_dict(seed=4))
    assert eng.is_roi_encoder and abs(eng.cond_scale - 0.8) < 1e-6
    eng.import_pyramid(_feats(g, f"s{S}_feat"), (192, 256))
    code = eng.codegen(torch.from_numpy(g[f"s{S}_boxes"])).cpu().numpy()
    np.testing.assert_allclose(code[:256], g[f"s{S}_cls_conv"].reshape(-1), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(code[256], g[f"s{S}_cls_bias"].reshape(-1)[0], atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_roi_encoder_classes_batched_equal_one_class_per_call(golden_dir, dtype):
    """Two classes of 5 shots in ONE batch (sylph_codegen_classes with the ROIEncoder): class 0 is the reference golden's support set,
    class 1 the same images in another order.  fp32: class 0 reproduces the golden code, and both codes are bit-identical to
    one-class-per-call runs (a class never meets another class's tokens)."""
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g7_roi_encoder.npz"))
    eng = _engine(dtype, _roienc_cfg())
    eng.load_state_dict(W.roi_encoder_state_dict(seed=4))
    feats = _feats(g, "s5_feat")
    boxes = torch.from_numpy(g["s5_boxes"])
    perm = torch.tensor([3, 0, 4, 1, 2])
    feats2 = [torch.cat([f, f[perm]], dim=0) for f in feats]
    boxes2 = torch.cat([boxes, boxes[perm]], dim=0)
    eng.import_pyramid(feats2, (192, 256))
    many = eng.codegen_classes(boxes2, 5).clone()
    assert tuple(many.shape) == (2, 257)
    singles = []
    for fs, bx in ((feats, boxes), ([f[perm] for f in feats], boxes[perm])):
        eng.import_pyramid(fs, (192, 256))
        singles.append(eng.codegen(bx).clone())
    if dtype == "f32":
        np.testing.assert_allclose(many[0, :256].cpu().numpy(), g["s5_cls_conv"].reshape(-1), atol=1e-3, rtol=1e-3)
        for k in range(2):
            assert torch.equal(many[k], singles[k]), f"class {k}: batched code differs from the one-class call"
    else:
        for k in range(2):
            assert F.cosine_similarity(many[k, :256], singles[k][:256], dim=0).item() > 0.9995
    assert F.cosine_similarity(many[0, :256], many[1, :256], dim=0).item() > 0.999  # same support set, another order


def test_roi_encoder_bf16_close(golden_dir):
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g7_roi_encoder.npz"))
    eng = _engine("bf16", _roienc_cfg())
    eng.load_state_dict(W.roi_encoder_state_dict(seed=4))
    eng.import_pyramid(_feats(g, "s5_feat"), (192, 256))
    code = eng.codegen(torch.from_numpy(g["s5_boxes"])).cpu()
    ref = torch.from_numpy(g["s5_cls_conv"].reshape(-1))
    assert F.cosine_similarity(code[:256], ref, dim=0).item() > 0.995


# --------------------------------------------------------------------------------- C3 / C4 shaped cases
def test_r101_backbone_matches_oracle_f32():
    """BASELINE config C4 backbone (MODEL.RESNETS.DEPTH 101) at small size."""
    from oracle import backbone as OB
    from sylph_amd import synthetic as W
    sd = W.backbone_state_dict(0, depth=101)
    imgs = W.synthetic_images(1, 96, 128, seed=8)
    eng = _engine("f32", _cfg(**{"MODEL.RESNETS.DEPTH": 101}))
    eng.load_state_dict(sd)
    eng.preprocess(imgs)
    eng.backbone()
    got = eng.export_pyramid()
    x, _ = OB.preprocess(imgs)
    ref = OB.backbone_fpn(x, sd, 101)
    for l in range(5):
        err = (got[l].cpu() - ref[l]).abs().max().item()
        assert err <= 1e-3 * max(1.0, ref[l].abs().max().item()), f"level {l}: {err}"


@pytest.mark.parametrize("N,thr,post", [(337, 0.02, 300), (866, 0.03, 300), (20, 0.011, 100)])
def test_many_class_decode_topk_matches_oracle(g1, N, thr, post):
    """Many-way decode (LVIS-sized N): > PRE_NMS_TOPK candidates on a level -> exact radix top-k, NMS, keep
    POST_NMS_TOPK(+ties).  Decode is checked on the SAME head outputs (exported), so every difference is decode."""
    from oracle import decode as OD
    from sylph_amd import synthetic as W
    cfg = _cfg(**{"MODEL.FCOS.INFERENCE_TH_TEST": thr, "MODEL.FCOS.POST_NMS_TOPK_TEST": post})
    eng = _engine("f32", cfg)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    codes = W.synthetic_codes(N, seed=77, scale=2.0)
    eng.head(codes["cls_conv"], codes["cls_bias"])
    lo, rg, ct, io = [[t.cpu() for t in ts] for ts in eng.export_head()]
    want = OD.predict_proposals(lo, rg, ct, io, pre_nms_thresh=thr, post_nms_topk=post)
    ncand_l0 = int((lo[0][0].sigmoid() > thr).sum())
    assert ncand_l0 > 1000, f"case does not exercise top-k ({ncand_l0})"
    got = eng.decode()
    for i, (w, g) in enumerate(zip(want, got)):
        w = OD.detector_postprocess(w, sizes[i], sizes[i][0], sizes[i][1])
        assert g["scores"].numel() == w["scores"].numel()
        np.testing.assert_array_equal(g["pred_classes"].cpu().numpy(), w["pred_classes"].numpy())
        np.testing.assert_array_equal(g["locations"].cpu().numpy(), w["locations"].numpy())
        np.testing.assert_allclose(g["scores"].cpu().numpy(), w["scores"].numpy(), atol=1e-5)
        np.testing.assert_allclose(g["pred_boxes"].cpu().numpy(), w["pred_boxes"].numpy(), atol=1e-3)


@pytest.mark.parametrize("N,thr,scale,bq,twc", [(866, 0.03, 2.0, ["ctrness"], False), (337, 0.02, 2.0, ["ctrness", "iou"], True),
                                                 (100, 0.05, 3.0, ["iou"], False), (33, 0.0, 1.0, ["ctrness"], False)])
def test_many_way_fused_scan_equals_unfused(g1, N, thr, scale, bq, twc):
    """bf16, more than 32 classes: the class-conditional conv and the score scan are ONE kernel (logits_scan_kernel), the logits
    never reach HBM.  Its candidates must be those of the unfused path (GroupNorm apply -> conv_igemm -> decode_scan_kernel):
    the detections of the fused step are compared bit for bit with a decode of the exported (unfused) logits, which are also
    decoded by the oracle."""
    from oracle import decode as OD
    from sylph_amd import synthetic as W
    cfg = _cfg(**{"MODEL.FCOS.INFERENCE_TH_TEST": thr, "MODEL.FCOS.POST_NMS_TOPK_TEST": 300, "MODEL.FCOS.BOX_QUALITY": bq,
                  "MODEL.FCOS.THRESH_WITH_CTR": twc})
    eng = _engine("bf16", cfg)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    codes = W.synthetic_codes(N, seed=77, scale=scale)
    eng.profile_enable(True); eng.profile_read()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    fused = eng.decode(max_out=5000)
    assert "logits_scan_kernel" in eng.profile_read()["kernels"], "the fused many-way head did not run"
    eng.profile_enable(False)
    lo, rg, ct, io = eng.export_head()  # runs the unfused conv on the same tower output
    eng.import_head(lo, rg, ct, io)     # ... and its logits go through decode_scan_kernel
    unfused = eng.decode(max_out=5000)
    want = OD.predict_proposals([t.cpu() for t in lo], [t.cpu() for t in rg], [t.cpu() for t in ct], [t.cpu() for t in io],
                                pre_nms_thresh=thr, post_nms_topk=300, thresh_with_ctr=twc, box_quality=bq)
    n_l0 = int((lo[0][0].float().sigmoid() > thr).sum())
    assert n_l0 > 100, f"too few candidates for a meaningful comparison ({n_l0})"
    for i, (f, u, w) in enumerate(zip(fused, unfused, want)):
        assert f["scores"].numel() == u["scores"].numel() > 0
        for k in ("pred_classes", "fpn_levels", "locations", "scores", "pred_boxes"):
            assert torch.equal(f[k], u[k]), f"image {i}: {k} differs between the fused and the unfused head"
        w = OD.detector_postprocess(w, sizes[i], sizes[i][0], sizes[i][1])
        assert f["scores"].numel() == w["scores"].numel()
        np.testing.assert_array_equal(f["pred_classes"].cpu().numpy(), w["pred_classes"].numpy())
        np.testing.assert_array_equal(f["locations"].cpu().numpy(), w["locations"].numpy())
        np.testing.assert_allclose(f["scores"].cpu().numpy(), w["scores"].numpy(), atol=1e-5)


def test_decode_topk_with_piled_up_scores_matches_oracle(g1):
    """Zero class codes: every class of a location has the SAME score, so the bin of the 1000-th largest holds far more
    candidates than the select kernel's on-chip list -> the whole-buffer radix select path, and the k boundary falls inside
    a run of exact ties (lower (location, class) index first, as the oracle's stable sort)."""
    from oracle import decode as OD
    from sylph_amd import synthetic as W
    N, thr, post = 800, 0.05, 300  # 320 locations x 800 classes <= 262 144: the buffer holds every score
    cfg = _cfg(**{"MODEL.FCOS.INFERENCE_TH_TEST": thr, "MODEL.FCOS.POST_NMS_TOPK_TEST": post})
    eng = _engine("f32", cfg)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.head(torch.zeros(N, 256, 1, 1), torch.zeros(N))
    lo, rg, ct, io = [[t.cpu() for t in ts] for ts in eng.export_head()]
    assert float(lo[0].abs().max()) == 0.0
    want = OD.predict_proposals(lo, rg, ct, io, pre_nms_thresh=thr, post_nms_topk=post)
    got = eng.decode(max_out=5000)  # the post-NMS keep takes every tie of the 300-th score: whole runs of classes
    for i, (w, g) in enumerate(zip(want, got)):
        w = OD.detector_postprocess(w, sizes[i], sizes[i][0], sizes[i][1])
        assert g["scores"].numel() == w["scores"].numel() > post
        np.testing.assert_array_equal(g["pred_classes"].cpu().numpy(), w["pred_classes"].numpy())
        np.testing.assert_array_equal(g["locations"].cpu().numpy(), w["locations"].numpy())
        np.testing.assert_allclose(g["scores"].cpu().numpy(), w["scores"].numpy(), atol=1e-5)


def test_candidate_overflow_fails_loudly(g1):
    from sylph_amd import synthetic as W
    eng = _engine("f32", _cfg(**{"MODEL.FCOS.INFERENCE_TH_TEST": 0.011}), cand_cap=64)
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    eng.import_pyramid(_feats(g1), (128, 160))
    codes = W.synthetic_codes(20, seed=50, scale=2.5)
    eng.head(codes["cls_conv"], codes["cls_bias"])
    with pytest.raises(RuntimeError, match="candidate capacity"):
        eng.decode()


def test_c3_shape_runs_bf16(full_sd):
    """BASELINE config C3 shape: 20-way, batch 16 queries (smaller images to keep the test short)."""
    from sylph_amd import synthetic as W
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(full_sd)
    q = W.synthetic_images(16, 256, 320, seed=12)
    codes = W.synthetic_codes(20, seed=13, scale=3.0)
    eng.preprocess(q)
    eng.backbone()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    dets = eng.decode()
    assert len(dets) == 16 and all(d["scores"].numel() > 0 for d in dets)
    assert all(int(d["pred_classes"].max()) < 20 for d in dets)


def test_c4_shape_runs_bf16_full_size():
    """BASELINE config C4 at full query size: R-101, 866 classes (LVIS), 300 detections, batch 2 of 800x1333.
    Exercises the production kernel selection (halo tiles, stem kernel, radix top-k over 866-wide score rows)."""
    from sylph_amd import synthetic as W
    cfg = _cfg(**{"MODEL.RESNETS.DEPTH": 101, "MODEL.FCOS.POST_NMS_TOPK_TEST": 300})
    eng = _engine("bf16", cfg)
    eng.load_state_dict(W.synthetic_state_dict(0, depth=101))
    q = W.synthetic_images(2, 800, 1333, seed=21)
    codes = W.synthetic_codes(866, seed=22, scale=1.5)  # ~ a few % of the 19.4 M scores per image pass the 0.05 threshold
    eng.preprocess(q)
    eng.backbone()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    dets = eng.decode()
    assert len(dets) == 2
    for d in dets:
        n = d["scores"].numel()
        assert 0 < n and int(d["pred_classes"].max()) < 866
        s = d["scores"].float().cpu()
        assert torch.all(s[:-1] >= s[1:]) or n == 1, "NMS output is sorted by score"
        b = d["pred_boxes"].float().cpu()
        assert torch.all(b[:, 2] >= b[:, 0]) and torch.all(b[:, 3] >= b[:, 1])
        assert float(b[:, 2].max()) <= 1333.0 + 1e-3 and float(b[:, 3].max()) <= 800.0 + 1e-3


def test_c4_full_size_f32_matches_oracle():
    """BASELINE config C4 against the ORACLE at full size: R-101-FPN, 866 classes, POST_NMS_TOPK 300, one 800x1333
    query, fp32 mode.  (1) logits / reg / ctrness / iou of all 22 400 locations x 866 classes within 1e-3 of the CPU oracle;
    (2) the oracle decoder on the HIP head outputs yields IDENTICAL candidates (exact ordinals: threshold over 19.4 M scores,
    per-level top-1000 of > 1000 candidates, class-aware NMS, top-300 + ties); (3) end to end >= 97 % of the oracle's detections
    with the same (level, location, class) and every differing one proved marginal."""
    from oracle import backbone as OB, decode as OD, head as OH
    from sylph_amd import synthetic as W
    if any(k.startswith(("SYLPH_CONV", "SYLPH_FUSE", "SYLPH_GN_FUSE")) for k in os.environ):
        pytest.skip("default kernel selection only")
    sd = W.synthetic_state_dict(0, depth=101)
    cfg = _cfg(**{"MODEL.RESNETS.DEPTH": 101, "MODEL.FCOS.POST_NMS_TOPK_TEST": 300})
    eng = _engine("f32", cfg)
    eng.load_state_dict(sd)
    q = W.synthetic_images(1, 800, 1333, seed=21)
    codes = W.synthetic_codes(866, seed=22, scale=1.5)
    x, sizes = OB.preprocess(q)
    with torch.no_grad():
        ref_head = OH.fcos_head(OB.backbone_fpn(x, sd, 101), sd, codes)
    eng.preprocess(q)
    eng.backbone()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    hip_head = [[t.cpu() for t in ts] for ts in eng.export_head()]
    got = eng.decode()[0]
    for name, hs, rs in zip(("logits", "reg", "ctrness", "iou"), hip_head, ref_head):  # (1)
        for l in range(5):
            err = (hs[l] - rs[l]).abs().max().item()
            assert err <= 1e-3, f"{name} level {l}: max err {err}"
    assert int((hip_head[0][0][0].sigmoid() > 0.05).sum()) > 1000, "the case must exercise the per-level top-k"
    wh = OD.detector_postprocess(OD.predict_proposals(*hip_head, post_nms_topk=300)[0], sizes[0], sizes[0][0], sizes[0][1])  # (2)
    assert got["scores"].numel() == wh["scores"].numel() >= 300
    np.testing.assert_array_equal(got["cand_index"].cpu().numpy(), _cand_ordinals(wh, 800, 1344, 866))
    np.testing.assert_allclose(got["scores"].cpu().numpy(), wh["scores"].numpy(), atol=1e-5)
    np.testing.assert_allclose(got["pred_boxes"].cpu().numpy(), wh["pred_boxes"].numpy(), atol=1e-3)
    wv = OD.detector_postprocess(OD.predict_proposals(*ref_head, post_nms_topk=300)[0], sizes[0], sizes[0][0], sizes[0][1])  # (3)
    ref_ord, hip_ord = _cand_ordinals(wv, 800, 1344, 866), got["cand_index"].cpu().numpy()
    hit = np.isin(ref_ord, hip_ord)
    print(f"C4 full-size fp32: {hit.sum()} of {hit.size} oracle detections reproduced exactly")
    assert hit.mean() >= 0.97, hit.mean()
    _prove_residue(ref_head, _keys_of_ordinals(ref_ord, 800, 1344, 866), hip_head, _keys_of_ordinals(hip_ord, 800, 1344, 866), 0,
                   "C4 full-size fp32", eps_val=1e-3, eps_iou=5e-3, post_nms_topk=300)


def test_c5_query_shape_runs_bf16():
    """BASELINE config C5 query geometry: 800x1200 queries (padded to 800x1216: level widths 152/76/38/19/10, not
    multiples of the 16-wide halo patches), 337 classes."""
    from sylph_amd import synthetic as W
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(W.synthetic_state_dict(0, depth=50))
    q = W.synthetic_images(3, 800, 1200, seed=31)
    codes = W.synthetic_codes(337, seed=32, scale=1.5)
    eng.preprocess(q)
    eng.backbone()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    dets = eng.decode()
    assert len(dets) == 3 and all(d["scores"].numel() > 0 for d in dets)
    pyr = eng.export_pyramid()
    assert [tuple(p.shape[-2:]) for p in pyr] == [(100, 152), (50, 76), (25, 38), (13, 19), (7, 10)]
    assert all(bool(torch.isfinite(p.float()).all()) for p in pyr)


# --------------------------------------------------------------------------------- round-2 additions
STRIDES = np.array([8, 16, 32, 64, 128])


def _assert_boxes(got, want, levels, what=""):
    """Boxes are location -+ reg * stride, so an error eps in the regression map is eps * stride pixels.  The north-star
    bound (1e-3 on the fp32 network outputs, checked directly on the reg maps by the head golden test) therefore
    reads |dbox| <= 1e-3 * stride; measured on MI355X: <= 2e-4 * stride (fp32 summation order of the 2304-term
    tower dot products, 4 layers deep).  We assert the measured bound with a small floor for the postprocess rescale."""
    got, want, levels = np.asarray(got, np.float64), np.asarray(want, np.float64), np.asarray(levels)
    tol = 2.5e-4 * STRIDES[levels][:, None] + 1e-3
    err = np.abs(got - want)
    assert (err <= tol).all(), f"{what} max box error {err.max():.4g} px; worst err/stride {(err / STRIDES[levels][:, None]).max():.3g}"


def test_normalisation_fused_into_the_stem_is_bit_identical(full_sd, monkeypatch):
    """bf16: sylph_preprocess leaves (x - mean) / std to the stem kernel (stem_pool_kernel<RAW>: fp32 planes -> normalised bf16 patch
    in LDS).  Ragged image sizes (zero padding in NORMALISED space, per image), against the separate preprocess pass: the res2..res5
    stage outputs and the pyramid must be bit-identical, and export_input still returns the normalised batch."""
    from sylph_amd import synthetic as W
    imgs = W.synthetic_images(3, 203, 333, seed=21)
    imgs = [imgs[0], imgs[1][:, :170, :301].contiguous(), imgs[2][:, :, :257].contiguous()]
    outs = []
    for knob in ("1", "0"):
        monkeypatch.setenv("SYLPH_FUSE_PREPROCESS", knob)
        eng = _engine("bf16", _cfg())
        eng.load_state_dict(full_sd)
        eng.preprocess(imgs)
        eng.backbone()
        outs.append(([t.clone() for t in eng.export_pyramid()], eng.export_input().clone()))
        eng.close()
    for a, b in zip(outs[0][0], outs[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].abs().max()) > 1.0


def test_stem_and_maxpool_kernels_bf16_vs_torch():
    """stem_conv_kernel (7x7 s2 p3, Cin 3, bf16) and maxpool_kernel directly, on sizes with ragged 8 x 16 tiles."""
    g = torch.Generator().manual_seed(11)
    for B, H, W in ((2, 64, 96), (1, 75, 118), (3, 33, 47)):
        x = _bf16_round(torch.randn(B, 3, H, W, generator=g) * 60.0)
        w = _bf16_round(torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5 / 60.0)
        scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
        eng = _engine("bf16")
        so, po = eng.stem_maxpool(x, w, scale, shift)
        ref = F.relu(F.conv2d(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        assert so.shape == ref.shape
        err = (so.cpu() - ref).abs().max().item()
        assert err <= 1e-2 * max(1.0, ref.abs().max().item()), f"stem {B}x{H}x{W}: max err {err}"
        # the max of bf16 values is exact: the pool must equal torch's pool of the kernel's own stem output bit for bit
        assert torch.equal(po.cpu(), F.max_pool2d(so.cpu(), 3, 2, 1)), f"maxpool {B}x{H}x{W}"


@pytest.mark.parametrize("k", [1, 2])
def test_cond_conv_block_matches_reference_golden(golden_dir, k):
    """CondConvBlock (ROIEncoder head, head_utils.py:121-162) with one and two 256-channel chunks vs the reference's
    own output (g5 ccb1 / ccb2).  NUM_CLS_CONVS = 0 makes the class-conditional conv read the imported features."""
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g5_reduce_condblock.npz"))
    cfg = _roienc_cfg()
    cfg.MODEL.FCOS.NUM_CLS_CONVS = 0
    cfg.MODEL.FCOS.NUM_BOX_CONVS = 0
    eng = _engine("f32", cfg)
    sd = W.head_state_dict(seed=1, num_classes=60)
    for i in range(k):
        sd[f"proposal_generator.fcos_head.cond_cls_logits.scales.{i}.scale"] = torch.tensor([1.0 / k])  # Scale init (head_utils.py:131-136)
    eng.load_state_dict(sd)
    feat = torch.from_numpy(g["ccb_feat"])
    eng.import_pyramid([feat] + [torch.zeros(2, 256, h, w) for h, w in ((3, 3), (2, 2), (1, 1), (1, 1))], (40, 48))
    eng.head(torch.from_numpy(g[f"ccb{k}_w"]), torch.from_numpy(g[f"ccb{k}_b"]))
    lo, _, _, _ = eng.export_head()
    np.testing.assert_allclose(lo[0].cpu().numpy(), g[f"ccb{k}_y"], atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("lvis", [False, True])
def test_codegen_10_shot_matches_reference_golden(golden_dir, lvis):
    """BASELINE config C3 support path: 10 shots of one class (mean over shots) + normalisation vs the reference."""
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g3b_codegen_s10.npz"))
    eng = _engine("f32", _cfg(lvis))
    eng.load_state_dict(W.codegen_state_dict(seed=2))
    eng.import_pyramid(_feats(g, "s10_feat"), (128, 160))
    code = eng.codegen(torch.from_numpy(g["s10_boxes"]))
    tag = f"{'lvis' if lvis else 'coco'}_s10"
    np.testing.assert_allclose(code[:256].cpu().numpy(), g[f"{tag}_cls_conv"].reshape(-1), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(code[256].item(), g[f"{tag}_cls_bias"].reshape(-1)[0], atol=1e-3, rtol=1e-3)
    out = eng.normalize_codes(code.reshape(1, 257).clone().contiguous()).cpu().numpy()
    np.testing.assert_allclose(out[0, :256], g[f"{tag}_norm_cls_conv"].reshape(-1), atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(out[0, 256], g[f"{tag}_norm_cls_bias"].reshape(-1)[0], atol=1e-4, rtol=1e-3)


def test_full_size_f32_matches_oracle(full_sd):
    """Two 800x1333 queries (ragged batch), 5-way, fp32 mode, DEFAULT kernel selection (no SYLPH_* knob): the production
    tile shapes, the XCD map with many M tiles per XCD, the fused top-down / shortcut paths -- against the CPU oracle.
    Three statements, each as strict as it can meaningfully be:
      (1) network outputs: logits / reg / ctrness / iou of all 22 400 locations within 1e-3 (the north-star bound);
      (2) decode + NMS + top-k + postprocess: the oracle decoder run on the HIP head outputs yields IDENTICAL candidates
          (exact index equality; scores 1e-5, boxes 1e-3 px);
      (3) end to end (two fp32 conv stacks with different summation orders feeding threshold / IoU > 0.6 / top-100
          decisions): >= 97 % of the oracle's detections are reproduced with the same (level, location, class), their
          scores within 1e-3 and boxes within the stride-relative bound -- and EVERY detection that only one side reports is
          PROVED marginal (_prove_residue): on the head outputs of the side that lacks it, it fails exactly one decision, by at
          most 1e-3 on cls x quality (the bound of (1)) or 5e-3 on an IoU (a 1e-3 x stride error of one box edge against a box
          at least a stride wide)."""
    from oracle import backbone as OB, decode as OD, head as OH
    from sylph_amd import synthetic as W
    if any(k.startswith(("SYLPH_CONV", "SYLPH_FUSE", "SYLPH_GN_FUSE")) for k in os.environ):
        pytest.skip("this test is about the DEFAULT kernel selection (forced-variant reruns set SYLPH_CONV_*)")
    eng = _engine("f32", _cfg())
    eng.load_state_dict(full_sd)
    q = W.synthetic_images(2, 800, 1333, seed=3)
    q[1] = q[1][:, :750, :1200].contiguous()  # ragged batch: second image padded on both axes
    codes = W.synthetic_codes(5, seed=4, scale=3.0)
    x, sizes = OB.preprocess(q)
    ref_head = OH.fcos_head(OB.backbone_fpn(x, full_sd, 50), full_sd, codes)
    assert eng.preprocess(q) == (800, 1344)
    eng.backbone()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    hip_head = [[t.cpu() for t in ts] for ts in eng.export_head()]
    got = eng.decode()
    for name, hs, rs in zip(("logits", "reg", "ctrness", "iou"), hip_head, ref_head):  # (1)
        for l in range(5):
            err = (hs[l] - rs[l]).abs().max().item()
            assert err <= 1e-3, f"{name} level {l}: max err {err}"
    want_on_hip = OD.predict_proposals(*hip_head)  # (2)
    want = OD.predict_proposals(*ref_head)
    for i in range(2):
        wh = OD.detector_postprocess(want_on_hip[i], sizes[i], sizes[i][0], sizes[i][1])
        gv = got[i]
        assert gv["scores"].numel() == wh["scores"].numel() >= 50
        np.testing.assert_array_equal(gv["cand_index"].cpu().numpy(), _cand_ordinals(wh, 800, 1344, 5))
        np.testing.assert_allclose(gv["scores"].cpu().numpy(), wh["scores"].numpy(), atol=1e-5)
        np.testing.assert_allclose(gv["pred_boxes"].cpu().numpy(), wh["pred_boxes"].numpy(), atol=1e-3)
        wv = OD.detector_postprocess(want[i], sizes[i], sizes[i][0], sizes[i][1])  # (3)
        ref_ord, hip_ord = _cand_ordinals(wv, 800, 1344, 5), gv["cand_index"].cpu().numpy()
        pos = {int(o): k for k, o in enumerate(hip_ord)}
        hit = np.array([o in pos for o in ref_ord.tolist()])
        print(f"full-size fp32 image {i}: {hit.sum()} of {hit.size} oracle detections reproduced exactly")
        assert hit.mean() >= 0.97, hit.mean()
        _prove_residue(ref_head, _keys_of_ordinals(ref_ord, 800, 1344, 5), hip_head, _keys_of_ordinals(hip_ord, 800, 1344, 5), i,
                       f"full-size fp32 image {i}", eps_val=1e-3, eps_iou=5e-3)
        sel = np.array([pos[int(o)] for o in ref_ord[hit].tolist()])
        np.testing.assert_allclose(gv["scores"].cpu().numpy()[sel], wv["scores"].numpy()[hit], atol=1e-3)
        _assert_boxes(gv["pred_boxes"].cpu().numpy()[sel], wv["pred_boxes"].numpy()[hit], wv["fpn_levels"].numpy()[hit], f"image {i}")


def _match_stats(got, want):
    """Fraction of oracle detections that have a HIP detection of the same class with IoU >= 0.9, and the largest score
    difference over those matches."""
    gb, gc, gs = got["pred_boxes"].float().cpu(), got["pred_classes"].cpu(), got["scores"].float().cpu()
    wb, wc, ws = want["pred_boxes"], want["pred_classes"], want["scores"]
    if wb.numel() == 0 or gb.numel() == 0:
        return 0.0, 0.0
    lt = torch.max(wb[:, None, :2], gb[None, :, :2])
    rb = torch.min(wb[:, None, 2:], gb[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    aw = (wb[:, 2] - wb[:, 0]) * (wb[:, 3] - wb[:, 1])
    ag = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
    iou = inter / (aw[:, None] + ag[None, :] - inter)
    iou = torch.where(wc[:, None] == gc[None, :], iou, torch.zeros_like(iou))
    best, idx = iou.max(dim=1)
    ok = best >= 0.9
    ds = (gs[idx] - ws).abs()[ok]
    return float(ok.float().mean()), float(ds.max()) if ds.numel() else 0.0


def test_c3_full_size_batch16_20way_bf16(full_sd):
    """BASELINE config C3 query path at full size: 20-way, batch 16 of 800x1333, bf16 (the production mode: halo / hpipe
    kernels, stem kernel).  Size-independent properties on all 16 images, and detection-level agreement with the fp32
    CPU oracle on one of them (same class + IoU >= 0.9 for >= 90 % of the oracle's detections; bf16 storage moves a few
    scores across the NMS / top-100 boundary)."""
    from oracle import episode as E
    from sylph_amd import synthetic as W
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(full_sd)
    q = W.synthetic_images(16, 800, 1333, seed=41)
    codes = W.synthetic_codes(20, seed=42, scale=3.0)
    assert eng.preprocess(q) == (800, 1344)
    eng.backbone()
    eng.head(codes["cls_conv"], codes["cls_bias"])
    dets = eng.decode()
    assert len(dets) == 16
    for d in dets:
        s, bx = d["scores"].float().cpu(), d["pred_boxes"].float().cpu()
        assert s.numel() > 0 and torch.isfinite(s).all() and (s[:-1] >= s[1:]).all()
        assert int(d["pred_classes"].max()) < 20
        assert (bx[:, 0] >= 0).all() and (bx[:, 2] <= 1333).all() and (bx[:, 1] >= 0).all() and (bx[:, 3] <= 800).all()
    want = E.forward_instances(q[5:6], codes, full_sd)[0]
    frac, dscore = _match_stats(dets[5], want)
    print(f"C3 bf16 vs fp32 oracle: matched {frac:.3f} of {want['scores'].numel()} detections, max |dscore| {dscore:.4f}")
    assert frac >= 0.9 and dscore <= 0.05, (frac, dscore)


def test_c3_support_path_10_shot_full_size_bf16(full_sd):
    """BASELINE config C3 support path at full size: one class, 10 support images of 800x1333, bf16, vs the fp32 CPU
    oracle (cosine of the un-normalised 256-d code and of the normalised one)."""
    from oracle import codegen as CG, episode as E
    from sylph_amd import synthetic as W
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(full_sd)
    sup = W.synthetic_images(10, 800, 1333, seed=61)
    boxes = W.synthetic_boxes(10, 800, 1333, seed=62)
    eng.preprocess(sup)
    eng.backbone()
    code = eng.codegen(boxes)
    ref = E.forward_class_code(sup, boxes, full_sd)
    cos = F.cosine_similarity(code[:256].float().cpu(), ref["cls_conv"].reshape(-1), dim=0).item()
    assert cos > 0.99, cos
    normed = eng.normalize_codes(code.reshape(1, 257).clone().contiguous()).cpu()
    rc, rb = CG.normalize_code(ref["cls_conv"], ref["cls_bias"], full_sd)
    assert F.cosine_similarity(normed[0, :256], rc.reshape(-1), dim=0).item() > 0.99
    assert abs(normed[0, 256].item() - rb.item()) < 5e-2


@pytest.mark.parametrize("tag", ["n5_t50", "n20_t50"])
def test_head_bf16_close_to_reference_golden(g1, tag):
    """The production (bf16) head -- conv_hpipe towers with the GroupNorm of layers 1-3 applied to the next conv's input
    halo in LDS, when selected / forced -- against the reference golden: bf16 storage of 8 activation tensors deep gives
    ~1e-2 relative deviations; bound measured on MI355X and asserted with margin."""
    from sylph_amd import synthetic as W
    eng = _engine("bf16", _cfg())
    eng.load_state_dict(W.head_state_dict(seed=1, num_classes=60))
    sizes = [tuple(int(v) for v in s) for s in g1["image_sizes"]]
    eng.import_pyramid(_feats(g1), (128, 160), sizes)
    eng.head(torch.from_numpy(g1[f"{tag}_cls_conv"]), torch.from_numpy(g1[f"{tag}_cls_bias"]))
    lo, rg, ct, io = eng.export_head()
    worst = 0.0
    for l in range(5):
        for got, want in ((lo[l], g1[f"{tag}_logits{l}"]), (rg[l], g1[f"reg{l}"]), (ct[l], g1[f"ctr{l}"]), (io[l], g1[f"iou{l}"])):
            err = np.abs(got.cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
            worst = max(worst, float(err))
    print(f"bf16 head vs reference golden ({tag}): worst relative deviation {worst:.4f}")
    assert worst < 4e-2, worst


@pytest.mark.parametrize("S", [2, 5])
def test_codegen_weight_and_scale_layers_match_reference_golden(golden_dir, g3, S):
    """CODE_GENERATOR.WEIGHT_LAYER / SCALE_LAYER on the HIP path (the three 1-channel heads as one stacked conv, softmax shot weights
    and the weighted pools in codegen_tail_kernel) against the reference's own outputs (g3c), fp32 mode <= 1e-3; then the
    normalisation with cls_weight_norm."""
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g3c_codegen_weight_scale.npz"))
    cfg = _cfg(**{"MODEL.META_LEARN.CODE_GENERATOR.WEIGHT_LAYER": ["", "", 1], "MODEL.META_LEARN.CODE_GENERATOR.SCALE_LAYER": ["", "", 1]})
    eng = _engine("f32", cfg)
    assert int(eng.sc.cg_has_weight) == 1 and int(eng.sc.cg_has_scale) == 1
    eng.load_state_dict(W.codegen_state_dict(seed=2, weight_scale_layers=True))
    eng.import_pyramid(_feats(g3, f"s{S}_feat"), (192, 256))
    code = eng.codegen(torch.from_numpy(g3[f"s{S}_boxes"]))
    wn = eng.codegen_weight_norm(1)
    np.testing.assert_allclose(code[:256].cpu().numpy(), g[f"s{S}_cls_conv"].reshape(-1), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(code[256].item(), g[f"s{S}_cls_bias"].reshape(-1)[0], atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(wn.cpu().numpy(), g[f"s{S}_cls_weight_norm"].reshape(-1), atol=1e-3, rtol=1e-3)
    i = {2: 0, 5: 1}[S]
    normed = eng.normalize_codes(code.reshape(1, 257).clone().contiguous(), wn).cpu().numpy()
    np.testing.assert_allclose(normed[0, :256], g[f"norm{i}_cls_conv"].reshape(-1), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(normed[0, 256], g[f"norm{i}_cls_bias"].reshape(-1)[0], atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tag,spec", [("mixed_tower", [["", "ReLU"], ["GN", ""], ["GN", "ReLU"]]), ("plain_tower", [["", ""]]), ("no_tower", [])])
def test_codegen_tower_variants_match_reference_golden(golden_dir, g3, tag, spec, dtype):
    """CODE_GENERATOR.TOWER_LAYERS entries other than ["GN", "ReLU"] (no norm / no activation / no tower, code_generator.py:648-688)
    on the HIP path against the reference's own outputs (g3d): fp32 <= 1e-3, bf16 to bf16 tolerance."""
    from sylph_amd import synthetic as W
    g = np.load(os.path.join(golden_dir, "g3d_codegen_variants.npz"))
    eng = _engine(dtype, _cfg(**{"MODEL.META_LEARN.CODE_GENERATOR.TOWER_LAYERS": spec}))
    assert int(eng.sc.cg_tower_layers) == len(spec)
    eng.load_state_dict(W.codegen_state_dict(seed=2, tower_spec=spec))
    for S in (2, 5):
        eng.import_pyramid(_feats(g3, f"s{S}_feat"), (192, 256))
        code = eng.codegen(torch.from_numpy(g3[f"s{S}_boxes"])).cpu().numpy()
        ref_c, ref_b = g[f"{tag}_s{S}_cls_conv"].reshape(-1), float(g[f"{tag}_s{S}_cls_bias"].reshape(-1)[0])
        tol = 1e-3 if dtype == "f32" else 3e-2
        assert np.abs(code[:256] - ref_c).max() <= tol * max(1.0, np.abs(ref_c).max()), (tag, S)
        assert abs(code[256] - ref_b) <= tol * max(1.0, abs(ref_b)), (tag, S)


def test_c4_support_path_r101_full_size_bf16():
    """BASELINE config C4 support path at full size: R-101 backbone, LVIS code-generator settings (BIAS_L2_NORM), two classes x 3
    support images of 800x1333 in ONE batch (sylph_codegen_classes), bf16, against the fp32 CPU oracle: cosine of the un-normalised
    and of the normalised 256-d codes, bias within 5e-2; and against the one-class-per-call path of the same engine."""
    from oracle import codegen as CG, episode as E
    from sylph_amd import synthetic as W
    sd = W.synthetic_state_dict(0, depth=101)
    eng = _engine("bf16", _cfg(True, **{"MODEL.RESNETS.DEPTH": 101}))
    eng.load_state_dict(sd)
    S, ncls = 3, 2
    sup = W.synthetic_images(S * ncls, 800, 1333, seed=71)
    boxes = W.synthetic_boxes(S * ncls, 800, 1333, seed=72)
    eng.preprocess(sup)
    eng.backbone()
    codes = eng.codegen_classes(boxes, S).clone()
    for k in range(ncls):
        eng.preprocess(sup[k * S:(k + 1) * S])
        eng.backbone()
        one = eng.codegen(boxes[k * S:(k + 1) * S])
        assert float((one[:256] - codes[k, :256]).abs().max()) <= 2e-2 * float(one[:256].abs().max())
    ref = E.forward_class_code(sup[:S], boxes[:S], sd, depth=101, bias_l2_norm=True)
    cos = F.cosine_similarity(codes[0, :256].float().cpu(), ref["cls_conv"].reshape(-1), dim=0).item()
    assert cos > 0.99, cos
    normed = eng.normalize_codes(codes[:1].clone().contiguous()).cpu()
    rc, rb = CG.normalize_code(ref["cls_conv"], ref["cls_bias"], sd)
    assert F.cosine_similarity(normed[0, :256], rc.reshape(-1), dim=0).item() > 0.99
    assert abs(normed[0, 256].item() - rb.item()) < 5e-2
