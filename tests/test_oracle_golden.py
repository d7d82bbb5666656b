"""Pin the oracle against golden vectors produced by the reference itself
(tests/golden/gen_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import codegen as CG
from oracle import decode as D
from oracle import episode as E
from oracle import head as H
from sylph_amd import synthetic as W

TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _checksum(sd, prefix):
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if k.startswith(prefix)))


def _feats(g, prefix="feat"):
    return [torch.from_numpy(g[f"{prefix}{l}_q8"].astype(np.float32) / 32.0) for l in range(5)]


@pytest.fixture(scope="module")
def g1(golden_dir):
    return _load(golden_dir, "g1_head_decode.npz")


@pytest.fixture(scope="module")
def head_sd(g1):
    sd = W.head_state_dict(seed=1, num_classes=60)
    assert abs(_checksum(sd, "proposal_generator") - float(g1["weights_checksum"])) < 1e-3
    return sd


@pytest.mark.parametrize("tag", ["n1_t50", "n5_t50", "n20_t50", "n20_t11"])
def test_head_logits_match_reference(g1, head_sd, tag):
    feats = _feats(g1)
    codes = {"cls_conv": torch.from_numpy(g1[f"{tag}_cls_conv"]), "cls_bias": torch.from_numpy(g1[f"{tag}_cls_bias"])}
    logits, regs, ctrs, ious = H.fcos_head(feats, head_sd, codes)
    for l in range(5):
        np.testing.assert_allclose(logits[l].numpy(), g1[f"{tag}_logits{l}"], atol=TOL, rtol=TOL)
    if tag == "n1_t50":
        for l in range(5):
            np.testing.assert_allclose(regs[l].numpy(), g1[f"reg{l}"], atol=TOL, rtol=TOL)
            np.testing.assert_allclose(ctrs[l].numpy(), g1[f"ctr{l}"], atol=TOL, rtol=TOL)
            np.testing.assert_allclose(ious[l].numpy(), g1[f"iou{l}"], atol=TOL, rtol=TOL)


@pytest.mark.parametrize("tag,thr", [("n1_t50", 0.05), ("n5_t50", 0.05), ("n20_t50", 0.05), ("n20_t11", 0.011)])
def test_decode_matches_reference(g1, head_sd, tag, thr):
    """Decode + NMS + top-k keep on the REFERENCE's own head outputs."""
    logits = [torch.from_numpy(g1[f"{tag}_logits{l}"]) for l in range(5)]
    regs = [torch.from_numpy(g1[f"reg{l}"]) for l in range(5)]
    ctrs = [torch.from_numpy(g1[f"ctr{l}"]) for l in range(5)]
    ious = [torch.from_numpy(g1[f"iou{l}"]) for l in range(5)]
    props = D.predict_proposals(logits, regs, ctrs, ious, pre_nms_thresh=thr)
    for i, p in enumerate(props):
        pre = f"{tag}_img{i}"
        assert p["scores"].numel() == int(g1[f"{tag}_count"][i])
        np.testing.assert_array_equal(p["pred_classes"].numpy(), g1[f"{pre}_pred_classes"])
        np.testing.assert_array_equal(p["fpn_levels"].numpy(), g1[f"{pre}_fpn_levels"])
        np.testing.assert_array_equal(p["locations"].numpy(), g1[f"{pre}_locations"])
        np.testing.assert_allclose(p["scores"].numpy(), g1[f"{pre}_scores"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(p["pred_boxes"].numpy(), g1[f"{pre}_pred_boxes"], atol=1e-4, rtol=1e-6)


HEAD_VARIANTS = [("share1", 1, "GN", False), ("nonorm", 0, "none", False), ("share2_nonorm", 2, "none", False), ("owd", 0, "GN", True)]


@pytest.mark.parametrize("tag,share,norm,owd", HEAD_VARIANTS)
def test_head_variants_match_reference(g1, golden_dir, tag, share, norm, owd):
    """MODEL.FCOS.NUM_SHARE_CONVS (shared tower, fcos.py:397,626), MODEL.FCOS.NORM "none" (fcos.py:72-122,399) and
    MODEL.PROPOSAL_GENERATOR.OWD (fcos_outputs.py:913-916) against the reference's own head outputs and proposals (g1c)."""
    g = _load(golden_dir, "g1c_head_variants.npz")
    sd = W.head_state_dict(seed=1, num_classes=60, num_share_convs=share, norm=norm)
    assert abs(_checksum(sd, "proposal_generator") - float(g[f"{tag}_weights_checksum"])) < 1e-3
    codes = {"cls_conv": torch.from_numpy(g["cls_conv"]), "cls_bias": torch.from_numpy(g["cls_bias"])}
    logits, regs, ctrs, ious = H.fcos_head(_feats(g1), sd, codes, num_share_convs=share, norm=norm)
    for l in range(5):
        np.testing.assert_allclose(logits[l].numpy(), g[f"{tag}_logits{l}"], atol=TOL, rtol=TOL)
        np.testing.assert_allclose(regs[l].numpy(), g[f"{tag}_reg{l}"], atol=TOL, rtol=TOL)
        np.testing.assert_allclose(ctrs[l].numpy(), g[f"{tag}_ctr{l}"], atol=TOL, rtol=TOL)
    ref = lambda k: [torch.from_numpy(g[f"{tag}_{k}{l}"]) for l in range(5)]
    props = D.predict_proposals(ref("logits"), ref("reg"), ref("ctr"), ious, owd=owd)
    for i, p in enumerate(props):
        pre = f"{tag}_img{i}"
        assert p["scores"].numel() == int(g[f"{tag}_count"][i])
        np.testing.assert_array_equal(p["pred_classes"].numpy(), g[f"{pre}_pred_classes"])
        np.testing.assert_array_equal(p["fpn_levels"].numpy(), g[f"{pre}_fpn_levels"])
        np.testing.assert_array_equal(p["locations"].numpy(), g[f"{pre}_locations"])
        np.testing.assert_allclose(p["scores"].numpy(), g[f"{pre}_scores"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(p["pred_boxes"].numpy(), g[f"{pre}_pred_boxes"], atol=1e-4, rtol=1e-6)


TOWER_DEPTH_CASES = [("c2b3", 2, 3, 0, "GN"), ("c1b4_share1", 1, 4, 1, "GN"), ("c3b1_nonorm", 3, 1, 0, "none"), ("c0b2", 0, 2, 0, "GN")]


@pytest.mark.parametrize("tag,nc,nb,share,norm", TOWER_DEPTH_CASES)
def test_unequal_tower_depths_match_reference(g1, golden_dir, tag, nc, nb, share, norm):
    """MODEL.FCOS.NUM_CLS_CONVS != NUM_BOX_CONVS (fcos.py:84-122), with / without the shared tower and GroupNorm, depth-0 cls tower:
    the reference's head outputs and proposals (g1e_tower_depths.npz)."""
    g = _load(golden_dir, "g1e_tower_depths.npz")
    sd = W.head_state_dict(seed=1, num_classes=60, num_share_convs=share, norm=norm, num_cls_convs=nc, num_box_convs=nb)
    assert abs(_checksum(sd, "proposal_generator") - float(g[f"{tag}_weights_checksum"])) < 1e-3
    codes = {"cls_conv": torch.from_numpy(g["cls_conv"]), "cls_bias": torch.from_numpy(g["cls_bias"])}
    logits, regs, ctrs, ious = H.fcos_head(_feats(g1), sd, codes, num_share_convs=share, norm=norm, num_cls_convs=nc, num_box_convs=nb)
    for l in range(5):
        for got, name in ((logits, "logits"), (regs, "reg"), (ctrs, "ctr"), (ious, "iou")):
            np.testing.assert_allclose(got[l].numpy(), g[f"{tag}_{name}{l}"], atol=TOL, rtol=TOL)
    ref = lambda k: [torch.from_numpy(g[f"{tag}_{k}{l}"]) for l in range(5)]
    props = D.predict_proposals(ref("logits"), ref("reg"), ref("ctr"), ref("iou"))
    for i, p in enumerate(props):
        pre = f"{tag}_img{i}"
        assert p["scores"].numel() == int(g[f"{tag}_count"][i])
        np.testing.assert_array_equal(p["pred_classes"].numpy(), g[f"{pre}_pred_classes"])
        np.testing.assert_array_equal(p["fpn_levels"].numpy(), g[f"{pre}_fpn_levels"])
        np.testing.assert_array_equal(p["locations"].numpy(), g[f"{pre}_locations"])
        np.testing.assert_allclose(p["scores"].numpy(), g[f"{pre}_scores"], atol=1e-6, rtol=1e-6)


OWD_CASES = [("ctr", ["ctrness"], False, 0.05, 0.6, 100), ("iou", ["iou"], False, 0.05, 0.6, 100), ("ctriou", ["ctrness", "iou"], False, 0.05, 0.6, 100),
             ("ctr_twc", ["ctrness"], True, 0.05, 0.6, 100), ("ctr_t20", ["ctrness"], False, 0.02, 0.6, 100),
             ("ctr_all", ["ctrness"], False, 0.05, 1.0, 1000), ("ctriou_all", ["ctrness", "iou"], False, 0.05, 1.0, 1000),
             ("ctr_top300", ["ctrness"], False, 0.05, 0.6, 300)]


def owd_head_state_dict():
    """The head of fixture g1d (tests/golden/gen_goldens.py::owd_head_state_dict): quality convs x4, biases -3 / -4.5."""
    sd = W.head_state_dict(seed=1, num_classes=60)
    p = "proposal_generator.fcos_head"
    for n, b in (("ctrness", -3.0), ("iou_overlap", -4.5)):
        sd[f"{p}.{n}.weight"] = sd[f"{p}.{n}.weight"] * 4.0
        sd[f"{p}.{n}.bias"] = torch.full_like(sd[f"{p}.{n}.bias"], b)
    return sd


@pytest.mark.parametrize("tag,bq,twc,thr,nms,post", OWD_CASES)
def test_owd_decode_matches_reference(g1, golden_dir, tag, bq, twc, thr, nms, post):
    """MODEL.PROPOSAL_GENERATOR.OWD on quality logits that straddle logit(0.05) on every level (g1d): the reference
    multiplies the all-ones class by the box quality BEFORE the threshold (fcos_outputs.py:937 `thresh_with_ctr or OWD`), so its
    candidates are the locations with quality > thresh.  Asserts the reference's PER-LEVEL candidate counts and its proposals; the
    `*_all` cases (NMS_TH 1, no post-NMS cut) expose the whole candidate set in the output."""
    g = _load(golden_dir, "g1d_owd_decode.npz")
    sd = owd_head_state_dict()
    assert abs(_checksum(sd, "proposal_generator") - float(g["weights_checksum"])) < 1e-3
    codes = {"cls_conv": torch.from_numpy(g["cls_conv"]), "cls_bias": torch.from_numpy(g["cls_bias"])}
    logits, regs, ctrs, ious = H.fcos_head(_feats(g1), sd, codes)
    for l in range(5):
        for got, name in ((regs, "reg"), (ctrs, "ctr"), (ious, "iou")):
            np.testing.assert_allclose(got[l].numpy(), g[f"{name}{l}"], atol=TOL, rtol=TOL)
        q = torch.from_numpy(g[f"ctr{l}"]).sigmoid()
        assert (q <= 0.05).float().mean() >= 0.3 and (q > 0.05).any()  # the fixture can see the order of threshold and multiply
    ref = lambda k: [torch.from_numpy(g[f"{k}{l}"]) for l in range(5)]
    strides = (8, 16, 32, 64, 128)
    for l in range(5):
        h, w = g[f"ctr{l}"].shape[-2:]
        per = D.decode_level(D.compute_locations(h, w, strides[l]), ref("logits")[l], ref("reg")[l] * strides[l], ref("ctr")[l], ref("iou")[l],
                             thr, 1000, twc, bq, owd=True)
        assert [p["scores"].numel() for p in per] == g[f"{tag}_level_counts"][:, l].tolist(), f"level {l} candidate count"
    props = D.predict_proposals(ref("logits"), ref("reg"), ref("ctr"), ref("iou"), pre_nms_thresh=thr, nms_thresh=nms, post_nms_topk=post,
                                thresh_with_ctr=twc, box_quality=bq, owd=True)
    for i, p in enumerate(props):
        pre = f"{tag}_img{i}"
        assert p["scores"].numel() == int(g[f"{tag}_count"][i])
        if tag.endswith("_all"):
            assert p["scores"].numel() == int(g[f"{tag}_level_counts"][i].sum())
        np.testing.assert_array_equal(p["pred_classes"].numpy(), g[f"{pre}_pred_classes"])
        np.testing.assert_array_equal(p["fpn_levels"].numpy(), g[f"{pre}_fpn_levels"])
        np.testing.assert_array_equal(p["locations"].numpy(), g[f"{pre}_locations"])
        np.testing.assert_allclose(p["scores"].numpy(), g[f"{pre}_scores"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(p["pred_boxes"].numpy(), g[f"{pre}_pred_boxes"], atol=1e-4, rtol=1e-6)


VARIANTS = [("iou", ["iou"], False), ("ctriou", ["ctrness", "iou"], False), ("ctr_twc", ["ctrness"], True),
            ("iou_twc", ["iou"], True), ("ctriou_twc", ["ctrness", "iou"], True)]


@pytest.mark.parametrize("tag,bq,twc", VARIANTS)
def test_decode_variants_match_reference(g1, golden_dir, tag, bq, twc):
    """MODEL.FCOS.BOX_QUALITY ["iou"] / ["ctrness","iou"] and THRESH_WITH_CTR (fcos_outputs.py:938-959) on the reference's own
    head outputs (g1's 5-way case) against the reference's proposals (g1b_decode_variants.npz)."""
    g = _load(golden_dir, "g1b_decode_variants.npz")
    logits = [torch.from_numpy(g1[f"n5_t50_logits{l}"]) for l in range(5)]
    regs = [torch.from_numpy(g1[f"reg{l}"]) for l in range(5)]
    ctrs = [torch.from_numpy(g1[f"ctr{l}"]) for l in range(5)]
    ious = [torch.from_numpy(g1[f"iou{l}"]) for l in range(5)]
    props = D.predict_proposals(logits, regs, ctrs, ious, box_quality=bq, thresh_with_ctr=twc)
    for i, p in enumerate(props):
        pre = f"{tag}_img{i}"
        assert p["scores"].numel() == int(g[f"{tag}_count"][i])
        np.testing.assert_array_equal(p["pred_classes"].numpy(), g[f"{pre}_pred_classes"])
        np.testing.assert_array_equal(p["fpn_levels"].numpy(), g[f"{pre}_fpn_levels"])
        np.testing.assert_array_equal(p["locations"].numpy(), g[f"{pre}_locations"])
        np.testing.assert_allclose(p["scores"].numpy(), g[f"{pre}_scores"], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(p["pred_boxes"].numpy(), g[f"{pre}_pred_boxes"], atol=1e-4, rtol=1e-6)


@pytest.fixture(scope="module")
def g3(golden_dir):
    return _load(golden_dir, "g3_codegen.npz")


@pytest.fixture(scope="module")
def cg_sd(g3):
    sd = W.codegen_state_dict(seed=2)
    assert abs(_checksum(sd, "code_generator") - float(g3["weights_checksum"])) < 1e-3
    return sd


@pytest.mark.parametrize("lvis", [False, True])
@pytest.mark.parametrize("S", [1, 2, 5])
def test_codegen_matches_reference(g3, cg_sd, lvis, S):
    feats = _feats(g3, f"s{S}_feat")
    boxes = torch.from_numpy(g3[f"s{S}_boxes"])
    code = CG.code_generator(feats, boxes, cg_sd, bias_l2_norm=lvis)
    tag = f"{'lvis' if lvis else 'coco'}_s{S}"
    assert code["cls_conv"].shape == (1, 256, 1, 1) and code["cls_bias"].shape == (1, 1, 1, 1)
    np.testing.assert_allclose(code["cls_conv"].numpy(), g3[f"{tag}_cls_conv"], atol=TOL, rtol=TOL)
    np.testing.assert_allclose(code["cls_bias"].numpy(), g3[f"{tag}_cls_bias"], atol=TOL, rtol=TOL)


@pytest.mark.parametrize("tagc", ["coco", "lvis"])
def test_normalize_and_format_match_reference(g3, cg_sd, tagc):
    codes = []
    for i, S in enumerate((1, 2, 5)):
        codes.append({"support_set_target": torch.tensor(i), "class_name": f"c{S}",
                      "class_code": {"cls_conv": torch.from_numpy(g3[f"{tagc}_s{S}_cls_conv"]),
                                     "cls_bias": torch.from_numpy(g3[f"{tagc}_s{S}_cls_bias"])}})
    normed = CG.forward_normalize_code(codes, cg_sd)
    for i, c in enumerate(normed):
        np.testing.assert_allclose(c["class_code"]["cls_conv"].numpy(), g3[f"{tagc}_norm{i}_cls_conv"], atol=TOL, rtol=TOL)
        np.testing.assert_allclose(c["class_code"]["cls_bias"].numpy(), g3[f"{tagc}_norm{i}_cls_bias"], atol=TOL, rtol=TOL)
        assert abs(float(c["class_code"]["cls_conv"].flatten().norm()) - 1.3) < 1e-4  # conv_scale * unit L2
    fm = E.format_class_codes_shared([normed[2], normed[0], normed[1]])
    assert fm["cls_conv"].shape == (3, 256, 1, 1) and fm["cls_bias"].shape == (3,)
    np.testing.assert_allclose(fm["cls_conv"].numpy(), g3[f"{tagc}_fmt_cls_conv"], atol=TOL, rtol=TOL)
    np.testing.assert_allclose(fm["cls_bias"].numpy(), g3[f"{tagc}_fmt_cls_bias"], atol=TOL, rtol=TOL)


def test_reduce_and_condblock_match_reference(golden_dir):
    g = _load(golden_dir, "g5_reduce_condblock.npz")
    chunks = []
    for i in range(5):
        cc = {k: torch.as_tensor(g[f"chunk{i}_{k}"]) for k in ("cls_conv", "cls_bias", "cls_weight_norm")}
        cc["acc_weight"] = float(g[f"chunk{i}_acc_weight"])
        chunks.append({"support_set_target": int(g[f"chunk{i}_cid"]), "class_name": "k", "class_code": cc})
    red = E.gather_class_code([chunks[:2], chunks[2:]], reduce=True)
    assert len(red) == 2
    for r in red:
        cid = r["support_set_target"]
        assert "acc_weight" not in r["class_code"]
        for k, v in r["class_code"].items():
            np.testing.assert_allclose(np.asarray(v), g[f"reduced{cid}_{k}"], atol=1e-6, rtol=1e-6)
    feat = torch.from_numpy(g["ccb_feat"])
    for k in (1, 2):
        y = H.cond_conv_block(feat, torch.from_numpy(g[f"ccb{k}_w"]), torch.from_numpy(g[f"ccb{k}_b"]))
        np.testing.assert_allclose(y.numpy(), g[f"ccb{k}_y"], atol=TOL, rtol=TOL)


@pytest.mark.parametrize("S", [2, 5])
def test_roi_encoder_matches_reference(golden_dir, S):
    """ROIEncoder code generator against the reference module's own output."""
    from oracle import roi_encoder as R
    g = _load(golden_dir, "g7_roi_encoder.npz")
    sd = W.roi_encoder_state_dict(seed=4)
    assert abs(_checksum(sd, "code_generator") - float(g["weights_checksum"])) < 1e-2
    out = R.roi_encoder(_feats(g, f"s{S}_feat"), torch.from_numpy(g[f"s{S}_boxes"]), sd, num_shots=S)
    assert out["cls_conv"].shape == (1, 256, 1, 1) and out["cls_bias"].shape == (1,)
    np.testing.assert_allclose(out["cls_conv"].numpy(), g[f"s{S}_cls_conv"], atol=5e-5, rtol=5e-5)
    np.testing.assert_allclose(out["cls_bias"].numpy(), g[f"s{S}_cls_bias"], atol=5e-5, rtol=5e-5)


TOWER_VARIANTS = [("mixed_tower", [["", "ReLU"], ["GN", ""], ["GN", "ReLU"]]), ("plain_tower", [["", ""]]), ("no_tower", [])]


@pytest.mark.parametrize("S", [2, 5])
@pytest.mark.parametrize("tag,spec", TOWER_VARIANTS)
def test_codegen_tower_variants_match_reference(golden_dir, g3, tag, spec, S):
    """CODE_GENERATOR.TOWER_LAYERS entries other than ["GN", "ReLU"] (code_generator.py:648-688) against the reference module (g3d)."""
    g = _load(golden_dir, "g3d_codegen_variants.npz")
    sd = W.codegen_state_dict(seed=2, tower_spec=spec)
    assert abs(_checksum(sd, "code_generator") - float(g[f"{tag}_weights_checksum"])) < 1e-3
    out = CG.code_generator(_feats(g3, f"s{S}_feat"), torch.from_numpy(g3[f"s{S}_boxes"]), sd, tower_spec=spec)
    np.testing.assert_allclose(out["cls_conv"].numpy(), g[f"{tag}_s{S}_cls_conv"], atol=TOL, rtol=TOL)
    np.testing.assert_allclose(out["cls_bias"].numpy(), g[f"{tag}_s{S}_cls_bias"], atol=TOL, rtol=TOL)


@pytest.mark.parametrize("S", [2, 5])
def test_codegen_weight_and_scale_layers_match_reference(golden_dir, g3, S):
    """CODE_GENERATOR.WEIGHT_LAYER (softmax shot weights) + SCALE_LAYER (cls_weight_norm), code_generator.py:583-645,766-829,969-999,
    against the reference module's own outputs on g3's inputs, then forward_normalize_code with the weight norm."""
    g = _load(golden_dir, "g3c_codegen_weight_scale.npz")
    sd = W.codegen_state_dict(seed=2, weight_scale_layers=True)
    assert abs(_checksum(sd, "code_generator") - float(g["weights_checksum"])) < 1e-3
    out = CG.code_generator(_feats(g3, f"s{S}_feat"), torch.from_numpy(g3[f"s{S}_boxes"]), sd, has_weight_layer=True, has_scale_layer=True)
    for k in ("cls_conv", "cls_bias", "cls_weight_norm"):
        np.testing.assert_allclose(out[k].numpy(), g[f"s{S}_{k}"], atol=TOL, rtol=TOL)
    i = {2: 0, 5: 1}[S]
    conv, bias = CG.normalize_code(out["cls_conv"], out["cls_bias"], sd, cls_weight_norm=out["cls_weight_norm"])
    np.testing.assert_allclose(conv.numpy(), g[f"norm{i}_cls_conv"], atol=TOL, rtol=TOL)
    np.testing.assert_allclose(bias.numpy(), g[f"norm{i}_cls_bias"], atol=TOL, rtol=TOL)
