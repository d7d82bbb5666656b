"""Generate golden vectors by running the REFERENCE's own Sylph modules (imported from
/root/reference through tests/golden/ref_shim.py) on seeded inputs.  Run in the build container:

    python tests/golden/gen_goldens.py

Writes tests/golden/*.npz (inputs + expected outputs; data only).  The reference never travels:
the GPU box replays these fixtures against the oracle and the HIP path.
Weights are NOT stored (too large); they are regenerated from sylph_amd.synthetic seeds and guarded by
a checksum stored in each fixture.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

ref_shim.install()

from sylph_amd import synthetic as W  # noqa: E402
from sylph_amd.config import get_default_cfg  # noqa: E402


def checksum(sd, prefix):
    return float(sum(v.double().abs().sum() for k, v in sorted(sd.items()) if k.startswith(prefix)))


def make_cfg(lvis=False):
    cfg = get_default_cfg()
    cfg.MODEL.DEVICE = "cpu"
    ml = cfg.MODEL.META_LEARN
    ml.EPISODIC_LEARNING = True
    cg = ml.CODE_GENERATOR
    cg.CONV_L2_NORM = True
    cg.TOWER_LAYERS = [["GN", "ReLU"], ["GN", "ReLU"]]
    cg.CLS_LAYER = ["", "", 1]
    cg.BIAS_LAYER = ["", "", 1]
    cfg.MODEL.PROPOSAL_GENERATOR.FREEZE_BBOX_BRANCH = True
    cfg.MODEL.FCOS.NUM_CLASSES = 60
    if lvis:
        cg.BIAS_L2_NORM = True
        cfg.MODEL.FCOS.NUM_CLASSES = 866
        cfg.MODEL.FCOS.POST_NMS_TOPK_TEST = 300
    return cfg


def load_prefixed(module, sd, prefix):
    sub = {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}
    missing, unexpected = module.load_state_dict(sub, strict=False)
    assert not unexpected, unexpected
    return missing


def feature_pyramid(B, h, w, seed, scale=1.0):
    """Features are multiples of 1/32 in [-4, 4) so fixtures can store them as int8."""
    g = torch.Generator().manual_seed(seed)
    feats = []
    for s in (8, 16, 32, 64, 128):
        hh, ww = -(-h // s), -(-w // s)
        q = torch.randint(-128, 128, (B, 256, hh, ww), generator=g, dtype=torch.int16)
        feats.append(q.float() / 32.0 * scale)
    return feats


def q8(t):
    q = (t * 32.0).round()
    assert torch.equal(q / 32.0, t) and q.abs().max() <= 128
    return q.to(torch.int8).numpy()


def inst_to_np(inst, prefix):
    out = {}
    for k, v in inst.get_fields().items():
        t = v.tensor if hasattr(v, "tensor") else v
        out[f"{prefix}_{k}"] = t.detach().numpy()
    return out


def gen_head_decode():
    from sylph.modeling.meta_fcos.fcos import MetaFCOS
    from ref_shim import ShapeSpec
    cfg = make_cfg()
    sd = W.head_state_dict(seed=1, num_classes=60)
    shapes = {f"p{l}": ShapeSpec(channels=256, stride=2 ** l) for l in range(3, 8)}
    model = MetaFCOS(cfg, shapes).eval()
    load_prefixed(model, sd, "proposal_generator")
    H, Wd, B = 128, 160, 2
    feats = feature_pyramid(B, H, Wd, seed=11)
    out = {"weights_checksum": checksum(sd, "proposal_generator"), "image_size": np.array([H, Wd])}
    for l, f in enumerate(feats):
        out[f"feat{l}_q8"] = q8(f)
    image_sizes = [(H, Wd - 7), (H - 5, Wd)]
    out["image_sizes"] = np.array(image_sizes)
    with torch.no_grad():
        for n, cscale, thr in ((1, 2.5, 0.05), (5, 2.0, 0.05), (20, 1.5, 0.05), (20, 2.5, 0.011)):
            tag = f"n{n}_t{int(thr * 1000)}"
            codes = W.synthetic_codes(n, seed=30 + n, scale=cscale)
            out[f"{tag}_cls_conv"] = codes["cls_conv"].numpy()
            out[f"{tag}_cls_bias"] = codes["cls_bias"].numpy()
            logits, reg, ctr, iou, _, _ = model.fcos_head(feats, None, False, codes)
            for l in range(5):
                out[f"{tag}_logits{l}"] = logits[l].numpy()
                if n == 1:
                    out[f"reg{l}"] = reg[l].numpy()
                    out[f"ctr{l}"] = ctr[l].numpy()
                    out[f"iou{l}"] = iou[l].numpy()
            model.fcos_outputs.pre_nms_thresh_test = thr
            locations = model.compute_locations(feats)
            props = model.fcos_outputs.predict_proposals(logits, reg, ctr, iou, locations, image_sizes, [])
            for i, p in enumerate(props):
                out.update(inst_to_np(p, f"{tag}_img{i}"))
            out[f"{tag}_count"] = np.array([len(p) for p in props])
            print("head/decode", tag, [len(p) for p in props])
    np.savez_compressed(os.path.join(HERE, "g1_head_decode.npz"), **out)


def gen_decode_variants():
    """predict_proposals under the other MODEL.FCOS.BOX_QUALITY / THRESH_WITH_CTR settings
    (sylph/modeling/meta_fcos/fcos_outputs.py:938-959) on the head outputs of g1's 5-way case: inputs are g1's
    `n5_t50_logits*`, `reg*`, `ctr*`, `iou*` (regenerated here, asserted equal), outputs are the reference's proposals.
    Separate file so that g1_head_decode.npz stays bit-stable."""
    from sylph.modeling.meta_fcos.fcos import MetaFCOS
    from ref_shim import ShapeSpec
    cfg = make_cfg()
    sd = W.head_state_dict(seed=1, num_classes=60)
    shapes = {f"p{l}": ShapeSpec(channels=256, stride=2 ** l) for l in range(3, 8)}
    model = MetaFCOS(cfg, shapes).eval()
    load_prefixed(model, sd, "proposal_generator")
    H, Wd, B = 128, 160, 2
    feats = feature_pyramid(B, H, Wd, seed=11)
    image_sizes = [(H, Wd - 7), (H - 5, Wd)]
    g1 = np.load(os.path.join(HERE, "g1_head_decode.npz"))
    out = {"weights_checksum": checksum(sd, "proposal_generator")}
    with torch.no_grad():
        codes = W.synthetic_codes(5, seed=35, scale=2.0)
        logits, reg, ctr, iou, _, _ = model.fcos_head(feats, None, False, codes)
        for l in range(5):
            assert np.array_equal(logits[l].numpy(), g1[f"n5_t50_logits{l}"]) and np.array_equal(iou[l].numpy(), g1[f"iou{l}"])
        locations = model.compute_locations(feats)
        for bq, twc, tag in ((["iou"], False, "iou"), (["ctrness", "iou"], False, "ctriou"), (["ctrness"], True, "ctr_twc"),
                             (["iou"], True, "iou_twc"), (["ctrness", "iou"], True, "ctriou_twc")):
            model.fcos_outputs.box_quality = bq
            model.fcos_outputs.thresh_with_ctr = twc
            model.fcos_outputs.pre_nms_thresh_test = 0.05
            props = model.fcos_outputs.predict_proposals(logits, reg, ctr, iou, locations, image_sizes, [])
            for i, p in enumerate(props):
                out.update(inst_to_np(p, f"{tag}_img{i}"))
            out[f"{tag}_count"] = np.array([len(p) for p in props])
            print("decode variant", tag, [len(p) for p in props])
    np.savez_compressed(os.path.join(HERE, "g1b_decode_variants.npz"), **out)


def gen_head_variants():
    """MetaFCOSHead / predict_proposals under config branches the five target yamls leave off (missing #3):
    MODEL.FCOS.NUM_SHARE_CONVS = 1 (shared tower, fcos.py:397,626), MODEL.FCOS.NORM = "none" (conv + ReLU towers, fcos.py:72-122,399) and
    MODEL.PROPOSAL_GENERATOR.OWD (one all-ones class, fcos_outputs.py:913-916) -- on g1's feature pyramid.  Separate file: g1 stays bit-stable."""
    from sylph.modeling.meta_fcos.fcos import MetaFCOS
    from ref_shim import ShapeSpec
    shapes = {f"p{l}": ShapeSpec(channels=256, stride=2 ** l) for l in range(3, 8)}
    H, Wd, B = 128, 160, 2
    feats = feature_pyramid(B, H, Wd, seed=11)
    image_sizes = [(H, Wd - 7), (H - 5, Wd)]
    codes = W.synthetic_codes(5, seed=35, scale=2.0)
    out = {"cls_conv": codes["cls_conv"].numpy(), "cls_bias": codes["cls_bias"].numpy(), "image_sizes": np.array(image_sizes)}
    with torch.no_grad():
        for tag, share, norm, owd in (("share1", 1, "GN", False), ("nonorm", 0, "none", False), ("share2_nonorm", 2, "none", False),
                                      ("owd", 0, "GN", True)):
            cfg = make_cfg()
            cfg.MODEL.FCOS.NUM_SHARE_CONVS = share
            cfg.MODEL.FCOS.NORM = norm
            cfg.MODEL.PROPOSAL_GENERATOR.OWD = owd
            sd = W.head_state_dict(seed=1, num_classes=60, num_share_convs=share, norm=norm)
            model = MetaFCOS(cfg, shapes).eval()
            load_prefixed(model, sd, "proposal_generator")
            out[f"{tag}_weights_checksum"] = checksum(sd, "proposal_generator")
            logits, reg, ctr, iou, _, _ = model.fcos_head(feats, None, False, codes)
            for l in range(5):
                out[f"{tag}_logits{l}"] = logits[l].numpy()
                out[f"{tag}_reg{l}"] = reg[l].numpy()
                out[f"{tag}_ctr{l}"] = ctr[l].numpy()
            locations = model.compute_locations(feats)
            props = model.fcos_outputs.predict_proposals(logits, reg, ctr, iou, locations, image_sizes, [])
            for i, p in enumerate(props):
                out.update(inst_to_np(p, f"{tag}_img{i}"))
            out[f"{tag}_count"] = np.array([len(p) for p in props])
            print("head variant", tag, [len(p) for p in props])
    np.savez_compressed(os.path.join(HERE, "g1c_head_variants.npz"), **out)


def owd_head_state_dict():
    """Head weights of the adversarial OWD fixture (g1d): g1's head with the centre-ness / IoU prediction convs scaled x4 and their
    biases moved to -3 / -4.5, so that sigmoid(quality) straddles the 0.05 threshold (logit -2.944) on every level.  Shared by the generator
    and the tests (the fixture stores a checksum, not the weights)."""
    sd = W.head_state_dict(seed=1, num_classes=60)
    p = "proposal_generator.fcos_head"
    for n, b in (("ctrness", -3.0), ("iou_overlap", -4.5)):
        sd[f"{p}.{n}.weight"] = sd[f"{p}.{n}.weight"] * 4.0
        sd[f"{p}.{n}.bias"] = torch.full_like(sd[f"{p}.{n}.bias"], b)
    return sd


def gen_owd_decode():
    """MODEL.PROPOSAL_GENERATOR.OWD decode on inputs that can SEE the order of the threshold and the quality multiply
    (fcos_outputs.py:937 `thresh_with_ctr or OWD`, :951 `not thresh_with_ctr and not OWD`): the quality logits straddle logit(0.05) on
    every level (the g1c `owd` case has every sigmoid(ctr) > 0.16 and cannot tell the two orders apart).
    Stores the reference's PER-LEVEL candidate counts (forward_for_single_feature_map, pre-NMS) besides the proposals, for the three
    BOX_QUALITY settings, with and without THRESH_WITH_CTR (which OWD makes irrelevant), and at a second threshold."""
    from sylph.modeling.meta_fcos.fcos import MetaFCOS
    from ref_shim import ShapeSpec
    shapes = {f"p{l}": ShapeSpec(channels=256, stride=2 ** l) for l in range(3, 8)}
    H, Wd, B = 128, 160, 2
    feats = feature_pyramid(B, H, Wd, seed=11)
    image_sizes = [(H, Wd - 7), (H - 5, Wd)]
    codes = W.synthetic_codes(5, seed=35, scale=2.0)
    sd = owd_head_state_dict()
    out = {"cls_conv": codes["cls_conv"].numpy(), "cls_bias": codes["cls_bias"].numpy(), "image_sizes": np.array(image_sizes),
           "weights_checksum": checksum(sd, "proposal_generator")}
    with torch.no_grad():
        cfg = make_cfg()
        cfg.MODEL.PROPOSAL_GENERATOR.OWD = True
        model = MetaFCOS(cfg, shapes).eval()
        load_prefixed(model, sd, "proposal_generator")
        logits, reg, ctr, iou, _, _ = model.fcos_head(feats, None, False, codes)
        for l in range(5):
            out[f"logits{l}"] = logits[l].numpy()
            out[f"reg{l}"] = reg[l].numpy()
            out[f"ctr{l}"] = ctr[l].numpy()
            out[f"iou{l}"] = iou[l].numpy()
            for name, t in (("ctr", ctr[l]), ("iou", iou[l])):
                below = float((t.sigmoid() <= 0.05).float().mean())
                print("owd quality", name, l, "fraction <= 0.05:", below)
                assert 0.3 <= below <= 0.95 or t[0].numel() <= 6, (name, l, below)  # both sides of the threshold populated on every level
        locations = model.compute_locations(feats)
        fo = model.fcos_outputs
        # (BOX_QUALITY, THRESH_WITH_CTR, INFERENCE_TH_TEST, NMS_TH, POST_NMS_TOPK_TEST, tag); `all`: nothing suppressed, nothing cut --
        # the proposals ARE the candidate set (the order of threshold and multiply is visible in the final output, not only in the counts);
        # `top300`: NMS as configured, no post-NMS cut
        for bq, twc, thr, nms, post, tag in ((["ctrness"], False, 0.05, 0.6, 100, "ctr"), (["iou"], False, 0.05, 0.6, 100, "iou"),
                                             (["ctrness", "iou"], False, 0.05, 0.6, 100, "ctriou"), (["ctrness"], True, 0.05, 0.6, 100, "ctr_twc"),
                                             (["ctrness"], False, 0.02, 0.6, 100, "ctr_t20"), (["ctrness"], False, 0.05, 1.0, 1000, "ctr_all"),
                                             (["ctrness", "iou"], False, 0.05, 1.0, 1000, "ctriou_all"), (["ctrness"], False, 0.05, 0.6, 300, "ctr_top300")):
            fo.box_quality, fo.thresh_with_ctr, fo.pre_nms_thresh_test, fo.nms_thresh, fo.post_nms_topk_test = bq, twc, thr, nms, post
            props = fo.predict_proposals(logits, reg, ctr, iou, locations, image_sizes, [])
            counts = np.zeros((B, 5), dtype=np.int64)
            for l in range(5):  # the same call predict_proposals makes per level (:800); pre_nms_thresh was set by the call above
                per = fo.forward_for_single_feature_map(locations[l], logits[l], reg[l] * fo.strides[l], ctr[l], iou[l], image_sizes, None)
                counts[:, l] = [len(p) for p in per]
            out[f"{tag}_level_counts"] = counts
            for i, p in enumerate(props):
                out.update(inst_to_np(p, f"{tag}_img{i}"))
            out[f"{tag}_count"] = np.array([len(p) for p in props])
            print("owd decode", tag, counts.tolist(), [len(p) for p in props])
    np.savez_compressed(os.path.join(HERE, "g1d_owd_decode.npz"), **out)


TOWER_DEPTH_CASES = (("c2b3", 2, 3, 0, "GN"), ("c1b4_share1", 1, 4, 1, "GN"), ("c3b1_nonorm", 3, 1, 0, "none"), ("c0b2", 0, 2, 0, "GN"))


def gen_tower_depths():
    """MetaFCOSHead with UNEQUAL tower depths (MODEL.FCOS.NUM_CLS_CONVS != NUM_BOX_CONVS, fcos.py:84-122: the two towers are built
    separately; the HIP path stacks them into one launch per layer only when the depths are equal, so unequal depths take other code)
    -- with and without the shared tower, with and without GroupNorm, and a cls tower of depth 0 (the class-conditional conv reads the
    pyramid / shared features directly).  Head outputs + proposals of the reference on g1's pyramid.  VERDICT r4 next #1 (re-audit of
    the round-4 branch additions with inputs that can see them)."""
    from sylph.modeling.meta_fcos.fcos import MetaFCOS
    from ref_shim import ShapeSpec
    shapes = {f"p{l}": ShapeSpec(channels=256, stride=2 ** l) for l in range(3, 8)}
    H, Wd, B = 128, 160, 2
    feats = feature_pyramid(B, H, Wd, seed=11)
    image_sizes = [(H, Wd - 7), (H - 5, Wd)]
    codes = W.synthetic_codes(5, seed=35, scale=2.0)
    out = {"cls_conv": codes["cls_conv"].numpy(), "cls_bias": codes["cls_bias"].numpy(), "image_sizes": np.array(image_sizes)}
    with torch.no_grad():
        for tag, nc, nb, share, norm in TOWER_DEPTH_CASES:
            cfg = make_cfg()
            cfg.MODEL.FCOS.NUM_CLS_CONVS, cfg.MODEL.FCOS.NUM_BOX_CONVS = nc, nb
            cfg.MODEL.FCOS.NUM_SHARE_CONVS, cfg.MODEL.FCOS.NORM = share, norm
            sd = W.head_state_dict(seed=1, num_classes=60, num_share_convs=share, norm=norm, num_cls_convs=nc, num_box_convs=nb)
            model = MetaFCOS(cfg, shapes).eval()
            missing = load_prefixed(model, sd, "proposal_generator")
            assert not [m for m in missing if "tower" in m], missing
            out[f"{tag}_weights_checksum"] = checksum(sd, "proposal_generator")
            logits, reg, ctr, iou, _, _ = model.fcos_head(feats, None, False, codes)
            for l in range(5):
                out[f"{tag}_logits{l}"] = logits[l].numpy()
                out[f"{tag}_reg{l}"] = reg[l].numpy()
                out[f"{tag}_ctr{l}"] = ctr[l].numpy()
                out[f"{tag}_iou{l}"] = iou[l].numpy()
            locations = model.compute_locations(feats)
            props = model.fcos_outputs.predict_proposals(logits, reg, ctr, iou, locations, image_sizes, [])
            for i, p in enumerate(props):
                out.update(inst_to_np(p, f"{tag}_img{i}"))
            out[f"{tag}_count"] = np.array([len(p) for p in props])
            print("tower depths", tag, [len(p) for p in props], float(logits[0].abs().max()))
    np.savez_compressed(os.path.join(HERE, "g1e_tower_depths.npz"), **out)


def gen_codegen():
    from sylph.modeling.code_generator.code_generator import CodeGenerator
    from ref_shim import Boxes, Instances
    import sylph.modeling.code_generator.code_generator as cgmod
    out = {}
    sd = W.codegen_state_dict(seed=2)
    out["weights_checksum"] = checksum(sd, "code_generator")
    H, Wd = 192, 256
    for lvis in (False, True):
        cfg = make_cfg(lvis)
        gen = CodeGenerator(cfg, 256, 5, cfg.MODEL.FCOS.FPN_STRIDES).eval()
        load_prefixed(gen, sd, "code_generator")
        tagc = "lvis" if lvis else "coco"
        all_codes = []
        for S in (1, 2, 5):
            feats = feature_pyramid(S, H, Wd, seed=100 + S)
            boxes = W.synthetic_boxes(S, H, Wd, seed=200 + S)
            if S == 5:  # exercise levels 3..6 and boundary-touching boxes
                boxes[0] = torch.tensor([4.0, 6.0, 60.0, 50.0])
                boxes[1] = torch.tensor([0.0, 0.0, 255.0, 191.0])
                boxes[2] = torch.tensor([60.5, 20.25, 250.75, 180.0])
            insts = []
            for i in range(S):
                it = Instances((H, Wd))
                it.gt_boxes = Boxes(boxes[i:i + 1])
                it.gt_classes = torch.tensor([3])
                insts.append(it)
            with torch.no_grad():
                code = gen(feats, insts)
            tag = f"{tagc}_s{S}"
            if not lvis:
                for l, f in enumerate(feats):
                    out[f"s{S}_feat{l}_q8"] = q8(f)
                out[f"s{S}_boxes"] = boxes.numpy()
            out[f"{tag}_cls_conv"] = code["cls_conv"].numpy()
            out[f"{tag}_cls_bias"] = code["cls_bias"].numpy()
            all_codes.append({"support_set_target": torch.tensor(len(all_codes)), "class_name": f"c{S}",
                              "class_code": {k: v.clone() for k, v in code.items()}})
            print("codegen", tag, code["cls_conv"].flatten()[:3], code["cls_bias"].flatten())
        # normalisation (run_type meta_learn_normalize_code) + formatting
        with torch.no_grad():
            normed = gen(None, None, cls_norm=True, class_codes=all_codes)
        for i, c in enumerate(normed):
            out[f"{tagc}_norm{i}_cls_conv"] = c["class_code"]["cls_conv"].numpy()
            out[f"{tagc}_norm{i}_cls_bias"] = c["class_code"]["cls_bias"].numpy()
        from sylph.evaluation.meta_learn_evaluation import format_class_codes_shared
        shuffled = [normed[2], normed[0], normed[1]]
        fm = format_class_codes_shared(shuffled, "cpu")
        out[f"{tagc}_fmt_cls_conv"] = fm["cls_conv"].numpy()
        out[f"{tagc}_fmt_cls_bias"] = fm["cls_bias"].numpy()
    np.savez_compressed(os.path.join(HERE, "g3_codegen.npz"), **out)


def gen_codegen_variants():
    """CodeGenerator branches the target yamls leave off (missing #3), on g3's S = 2 / 5 inputs:
    CODE_GENERATOR.TOWER_LAYERS with entries other than ["GN", "ReLU"] (no norm / no activation / no tower at all,
    code_generator.py:648-688).  ROI_BOX.FPN_MULTILEVEL_FEATURE is not here because the reference cannot run it: CodeGeneratorHead
    builds detectron2's ROIPooler (:26,343), whose output is ONE tensor, so :943 iterates over its batch dimension and GroupNorm
    fails on the unbatched (256, 7, 7) slices."""
    from sylph.modeling.code_generator.code_generator import CodeGenerator
    from ref_shim import Boxes, Instances
    out = {}
    H, Wd = 192, 256
    g3 = np.load(os.path.join(HERE, "g3_codegen.npz"))
    for tag, spec, multi in (("mixed_tower", [["", "ReLU"], ["GN", ""], ["GN", "ReLU"]], False), ("plain_tower", [["", ""]], False),
                             ("no_tower", [], False)):
        cfg = make_cfg(False)
        cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
        cg.TOWER_LAYERS = spec
        cg.ROI_BOX.FPN_MULTILEVEL_FEATURE = multi
        sd = W.codegen_state_dict(seed=2, tower_spec=spec)
        out[f"{tag}_weights_checksum"] = checksum(sd, "code_generator")
        gen = CodeGenerator(cfg, 256, 5, cfg.MODEL.FCOS.FPN_STRIDES).eval()
        load_prefixed(gen, sd, "code_generator")
        for S in (2, 5):
            feats = [torch.from_numpy(g3[f"s{S}_feat{l}_q8"].astype(np.float32) / 32.0) for l in range(5)]
            boxes = torch.from_numpy(g3[f"s{S}_boxes"])
            insts = []
            for i in range(S):
                it = Instances((H, Wd))
                it.gt_boxes = Boxes(boxes[i:i + 1])
                it.gt_classes = torch.tensor([3])
                insts.append(it)
            with torch.no_grad():
                code = gen(feats, insts)
            out[f"{tag}_s{S}_cls_conv"] = code["cls_conv"].numpy()
            out[f"{tag}_s{S}_cls_bias"] = code["cls_bias"].numpy()
            print("codegen variant", tag, S, code["cls_conv"].flatten()[:3], code["cls_bias"].flatten())
    np.savez_compressed(os.path.join(HERE, "g3d_codegen_variants.npz"), **out)


def gen_codegen_weight_scale():
    """CODE_GENERATOR.WEIGHT_LAYER (softmax shot weights) and SCALE_LAYER (cls_weight_norm) of the reference's CodeGenerator
    (code_generator.py:583-645,766-829,969-999) on g3's S = 2 / 5 inputs, then forward_normalize_code with the weight norm."""
    from sylph.modeling.code_generator.code_generator import CodeGenerator
    from ref_shim import Boxes, Instances
    out = {}
    sd = W.codegen_state_dict(seed=2, weight_scale_layers=True)
    out["weights_checksum"] = checksum(sd, "code_generator")
    H, Wd = 192, 256
    cfg = make_cfg(False)
    cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
    cg.WEIGHT_LAYER = ["", "", 1]
    cg.SCALE_LAYER = ["", "", 1]
    gen = CodeGenerator(cfg, 256, 5, cfg.MODEL.FCOS.FPN_STRIDES).eval()
    missing = load_prefixed(gen, sd, "code_generator")
    assert not [m for m in missing if "support_set_cls" in m], missing
    g3 = np.load(os.path.join(HERE, "g3_codegen.npz"))
    recs = []
    for S in (2, 5):
        feats = feature_pyramid(S, H, Wd, seed=100 + S)
        for l, f in enumerate(feats):
            assert np.array_equal(q8(f), g3[f"s{S}_feat{l}_q8"])  # same inputs as g3 (not stored again)
        boxes = torch.from_numpy(g3[f"s{S}_boxes"])
        insts = []
        for i in range(S):
            it = Instances((H, Wd))
            it.gt_boxes = Boxes(boxes[i:i + 1])
            it.gt_classes = torch.tensor([3])
            insts.append(it)
        with torch.no_grad():
            code = gen(feats, insts)
        for k in ("cls_conv", "cls_bias", "cls_weight_norm"):
            out[f"s{S}_{k}"] = code[k].numpy()
        recs.append({"support_set_target": torch.tensor(len(recs)), "class_name": f"c{S}", "class_code": {k: v.clone() for k, v in code.items()}})
        print("codegen weight/scale", S, code["cls_conv"].flatten()[:3], code["cls_bias"].flatten(), code["cls_weight_norm"].flatten())
    with torch.no_grad():
        normed = gen(None, None, cls_norm=True, class_codes=recs)
    for i, c in enumerate(normed):
        out[f"norm{i}_cls_conv"] = c["class_code"]["cls_conv"].numpy()
        out[f"norm{i}_cls_bias"] = c["class_code"]["cls_bias"].numpy()
    np.savez_compressed(os.path.join(HERE, "g3c_codegen_weight_scale.npz"), **out)


def gen_codegen_s10():
    """BASELINE config C3 support path: 10 shots of one class (mean over shots, code_generator.py:766-829), COCO and
    LVIS (BIAS_L2_NORM) settings, boxes on levels 3..6.  Separate file so that g3_codegen.npz stays bit-stable."""
    from sylph.modeling.code_generator.code_generator import CodeGenerator
    from ref_shim import Boxes, Instances
    out = {}
    sd = W.codegen_state_dict(seed=2)
    out["weights_checksum"] = checksum(sd, "code_generator")
    H, Wd, S = 128, 160, 10
    feats = feature_pyramid(S, H, Wd, seed=110)
    boxes = W.synthetic_boxes(S, H, Wd, seed=210)
    boxes[0] = torch.tensor([2.0, 3.0, 30.0, 28.0])       # level 3
    boxes[1] = torch.tensor([0.0, 0.0, 159.0, 127.0])     # whole image -> level 3 (sqrt(area) < 224)
    boxes[2] = torch.tensor([10.5, 20.25, 150.75, 120.0])
    boxes[3] = torch.tensor([100.0, 90.0, 160.0, 128.0])  # touches the bottom-right corner
    for l, f in enumerate(feats):
        out[f"s{S}_feat{l}_q8"] = q8(f)
    out[f"s{S}_boxes"] = boxes.numpy()
    for lvis in (False, True):
        cfg = make_cfg(lvis)
        gen = CodeGenerator(cfg, 256, 5, cfg.MODEL.FCOS.FPN_STRIDES).eval()
        load_prefixed(gen, sd, "code_generator")
        insts = []
        for i in range(S):
            it = Instances((H, Wd))
            it.gt_boxes = Boxes(boxes[i:i + 1])
            it.gt_classes = torch.tensor([3])
            insts.append(it)
        with torch.no_grad():
            code = gen(feats, insts)
            tag = f"{'lvis' if lvis else 'coco'}_s{S}"
            out[f"{tag}_cls_conv"] = code["cls_conv"].numpy()
            out[f"{tag}_cls_bias"] = code["cls_bias"].numpy()
            rec = [{"support_set_target": torch.tensor(0), "class_name": "c10", "class_code": {k: v.clone() for k, v in code.items()}}]
            normed = gen(None, None, cls_norm=True, class_codes=rec)
            out[f"{tag}_norm_cls_conv"] = normed[0]["class_code"]["cls_conv"].numpy()
            out[f"{tag}_norm_cls_bias"] = normed[0]["class_code"]["cls_bias"].numpy()
        print("codegen", tag, code["cls_conv"].flatten()[:3], code["cls_bias"].flatten())
    np.savez_compressed(os.path.join(HERE, "g3b_codegen_s10.npz"), **out)


def gen_reduce_condblock():
    from sylph.modeling.code_generator.utils import reduce_class_code
    from sylph.modeling.meta_fcos.head_utils import CondConvBlock
    import sylph.modeling.code_generator.utils as u
    import logging
    out = {}
    g = torch.Generator().manual_seed(7)
    chunks = []
    # class 0: three chunks with weights summing to 1; class 1: two chunks summing to 0.7
    for cid, wts in ((0, (0.5, 0.3, 0.2)), (1, (0.4, 0.3))):
        for w in wts:
            conv = torch.randn(1, 256, 1, 1, generator=g)
            bias = torch.randn(1, 1, 1, 1, generator=g)
            wn = torch.randn(1, 1, 1, 1, generator=g)
            chunks.append({"support_set_target": cid, "class_name": f"k{cid}",
                           "class_code": {"cls_conv": conv * w, "cls_bias": bias * w,
                                          "cls_weight_norm": wn * w, "acc_weight": w}})
    for i, c in enumerate(chunks):
        out[f"chunk{i}_cid"] = np.array(c["support_set_target"])
        for k, v in c["class_code"].items():
            out[f"chunk{i}_{k}"] = np.asarray(v)
    import copy
    red = reduce_class_code(copy.deepcopy(chunks))
    for r in red:
        cid = r["support_set_target"]
        for k, v in r["class_code"].items():
            out[f"reduced{cid}_{k}"] = np.asarray(v)
    # CondConvBlock 256 and 512 channels
    feat = torch.randn(2, 256, 5, 6, generator=g)
    for k in (1, 2):
        blk = CondConvBlock(padding=0, weight_len=256 * k)
        w = torch.randn(7, 256 * k, 1, 1, generator=g) * 0.1
        b = torch.randn(7, generator=g)
        with torch.no_grad():
            y = blk(feat, w, b)
        out[f"ccb{k}_w"], out[f"ccb{k}_b"], out[f"ccb{k}_y"] = w.numpy(), b.numpy(), y.numpy()
    out["ccb_feat"] = feat.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_reduce_condblock.npz"), **out)


def gen_roi_encoder():
    """ROIEncoder (LVIS ROI-encoder yaml dims) on seeded pyramids; S = EVAL_SHOT shots of one class."""
    from sylph.modeling.code_generator.roi_encoder import ROIEncoder
    from ref_shim import Boxes, Instances
    out = {}
    sd = W.roi_encoder_state_dict(seed=4)
    out["weights_checksum"] = checksum(sd, "code_generator")
    H, Wd = 192, 256
    for S in (2, 5):
        cfg = make_cfg(True)
        cg = cfg.MODEL.META_LEARN.CODE_GENERATOR
        cg.NAME = "ROIEncoder"
        cg.TOKENIZER.NUM_CONV, cg.TOKENIZER.CONV_DIM, cg.TOKENIZER.NORM = 2, 256, "GN"
        cg.TOKENIZER.NUM_FC, cg.TOKENIZER.FC_DIM = 2, 256
        cg.TRANSFORMER_ENCODER.LAYERS, cg.TRANSFORMER_ENCODER.HEADS = 2, 8
        cg.HEAD.NUM_FC, cg.HEAD.FC_DIM, cg.HEAD.OUTPUT_DIM = 2, 512, 256
        cfg.MODEL.META_LEARN.EVAL_SHOT = S
        enc = ROIEncoder(cfg, 256, 5, cfg.MODEL.FCOS.FPN_STRIDES).eval()
        missing = load_prefixed(enc, sd, "code_generator")
        assert not missing, missing
        feats = feature_pyramid(S, H, Wd, seed=300 + S)
        boxes = W.synthetic_boxes(S, H, Wd, seed=400 + S)
        insts = []
        for i in range(S):
            it = Instances((H, Wd))
            it.gt_boxes = Boxes(boxes[i:i + 1])
            it.gt_classes = torch.tensor([7])
            insts.append(it)
        with torch.no_grad():
            code = enc(feats, insts)
        for l, f in enumerate(feats):
            out[f"s{S}_feat{l}_q8"] = q8(f)
        out[f"s{S}_boxes"] = boxes.numpy()
        out[f"s{S}_cls_conv"] = code["cls_conv"].numpy()
        out[f"s{S}_cls_bias"] = code["cls_bias"].numpy()
        print("roi_encoder", S, code["cls_conv"].flatten()[:3], code["cls_bias"])
    np.savez_compressed(os.path.join(HERE, "g7_roi_encoder.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    np.random.seed(0)
    only = sys.argv[1:]  # e.g. `gen_goldens.py gen_codegen_s10` regenerates one fixture
    for fn in (gen_head_decode, gen_decode_variants, gen_head_variants, gen_owd_decode, gen_tower_depths, gen_codegen, gen_codegen_variants, gen_codegen_weight_scale, gen_codegen_s10, gen_reduce_condblock, gen_roi_encoder):
        if not only or fn.__name__ in only:
            fn()
    print("done")
