"""GPU tests of the reference-API mirror: MetaFCOSRunner episode flow, model(run_type=...) contract,
SylphPredictor with on-disk class codes -- against the CPU oracle (fp32 mode)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STRIDES = np.array([8, 16, 32, 64, 128])


def _assert_boxes(got, want, levels):
    """|dbox| <= 2.5e-4 * stride + 1e-3 px: a box is location -+ reg * stride, see tests/test_hip_parity.py::_assert_boxes."""
    got, want, levels = np.asarray(got, np.float64), np.asarray(want, np.float64), np.asarray(levels)
    err = np.abs(got - want)
    assert (err <= 2.5e-4 * STRIDES[levels][:, None] + 1e-3).all(), f"max box error {err.max():.4g} px"


def _cfg():
    from sylph_amd.runner import MetaFCOSRunner, create_cfg
    r = MetaFCOSRunner()
    return r, create_cfg(r.get_default_cfg(), "sylph://COCO-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml")


@pytest.fixture(scope="module")
def sd():
    from sylph_amd import synthetic as W
    return W.synthetic_state_dict(0, depth=50)


@pytest.fixture(scope="module", params=["f32", "f32s"])
def model(sd, request):
    """The two modes that carry the north-star's 1e-3 / identical-index statement: exact-fp32 MFMA and split-bf16 (fp32 storage, three
    bf16 MFMAs per product)."""
    runner, cfg = _cfg()
    m = runner.build_model(cfg, dtype=request.param)
    m.load_state_dict(sd)
    m.eval()
    return m


class _Collect:
    def reset(self):
        self.out = []

    def process(self, inputs, outputs):
        for i, o in zip(inputs, outputs):
            self.out.append((i["image_id"], o["instances"]))

    def evaluate(self):
        return {"n": len(self.out)}


@pytest.mark.parametrize("ways,shots,nq,bs", [(3, 2, 4, 2), (5, 1, 2, 1)], ids=["3way2shot_4q", "C1_5way1shot_2q"])
def test_runner_episode_matches_oracle(model, sd, ways, shots, nq, bs):
    """One episode through MetaFCOSRunner with the COCO Meta-FCOS-finetune.yaml against the fp32 oracle.  `C1_5way1shot_2q` is
    BASELINE.json configs[0] to the letter: R-50-FPN, 5-way 1-shot, 2 query images, query batch 1 (the reference's loop B,
    meta_learn_evaluation.py:421-426)."""
    from oracle import codegen as CG, episode as E
    from sylph_amd.data import SyntheticQueryLoader, SyntheticSupportSetLoader
    runner, cfg = _cfg()
    sup = SyntheticSupportSetLoader(ways, shots, 128, 160, seed=3)
    qry = SyntheticQueryLoader(nq, 120, 152, batch_size=bs, seed=4)
    ev = _Collect()
    res, codes = runner._do_test_meta_learning(cfg, model, sup, qry, ev, num_classes=ways)
    assert res == {"n": nq} and codes["cls_conv"].shape == (ways, 256, 1, 1) and codes["cls_bias"].shape == (ways,)
    # oracle: same support items -> codes
    recs = []
    for item in sup:
        it = item[0]
        imgs = [r["image"].cpu() for r in it["support_set"]]
        boxes = torch.cat([r["instances"].gt_boxes.tensor for r in it["support_set"]])
        recs.append({"support_set_target": it["support_set_target"], "class_name": it["class_name"],
                     "class_code": E.forward_class_code(imgs, boxes, sd)})
    ref = E.format_class_codes_shared(CG.forward_normalize_code(recs, sd))
    np.testing.assert_allclose(codes["cls_conv"].cpu().numpy(), ref["cls_conv"].numpy(), atol=1e-3)
    np.testing.assert_allclose(codes["cls_bias"].cpu().numpy(), ref["cls_bias"].numpy(), atol=1e-3)
    # queries with boosted codes so detections exist; same codes on both sides
    boosted = {"cls_conv": ref["cls_conv"] * 3.0, "cls_bias": ref["cls_bias"]}
    for batch in qry:
        got = model(batch, class_code={k: v.cuda() for k, v in boosted.items()}, run_type="meta_learn_test_instance")
        want = E.forward_instances([b["image"].cpu() for b in batch], boosted, sd)
        for g, w in zip(got, want):
            inst = g["instances"]
            assert inst.image_size == (120, 152) and len(inst) == w["scores"].numel() and len(inst) > 0
            perm = _order(model, inst, w)
            np.testing.assert_array_equal(inst.pred_classes.cpu().numpy(), w["pred_classes"].numpy()[perm])
            np.testing.assert_array_equal(inst.fpn_levels.cpu().numpy(), w["fpn_levels"].numpy()[perm])
            np.testing.assert_array_equal(inst.locations.cpu().numpy(), w["locations"].numpy()[perm])
            np.testing.assert_allclose(inst.scores.cpu().numpy(), w["scores"].numpy()[perm], atol=1e-3)
            _assert_boxes(inst.pred_boxes.tensor.cpu().numpy(), w["pred_boxes"].numpy()[perm], w["fpn_levels"].numpy()[perm])


def _order(model, inst, w):
    """Position in the oracle's list of every HIP detection.  Exact-fp32 mode: the identity (the arrays must be equal element by
    element).  Split-bf16 mode (2^-17 relative per product): the same SET of (level, location, class) triples, and two detections may
    trade places only where the oracle's own scores are within 1e-4 of each other (the list is sorted by score)."""
    n = len(inst)
    if model.engine.dtype != "f32s":
        return np.arange(n)
    key = lambda lv, loc, cl: [(int(a), float(b[0]), float(b[1]), int(c)) for a, b, c in zip(lv, loc, cl)]
    gk = key(inst.fpn_levels.cpu().numpy(), inst.locations.cpu().numpy(), inst.pred_classes.cpu().numpy())
    wk = key(w["fpn_levels"].numpy(), w["locations"].numpy(), w["pred_classes"].numpy())
    pos = {k: i for i, k in enumerate(wk)}
    assert len(pos) == n and set(gk) == set(wk), "different detection sets"
    perm = np.array([pos[k] for k in gk])
    moved = np.nonzero(perm != np.arange(n))[0]
    if moved.size:
        ws = w["scores"].numpy()
        assert np.abs(ws[perm[moved]] - ws[moved]).max() <= 1e-4
    return perm


def test_model_contract_errors(model):
    from sylph_amd.structures import Boxes, Instances
    empty = Instances((64, 64))
    empty.gt_boxes = Boxes(torch.zeros(0, 4))
    empty.gt_classes = torch.zeros(0, dtype=torch.long)
    item = [{"support_set": [{"image": torch.zeros(3, 64, 64), "instances": empty}],
             "support_set_target": torch.tensor(0), "class_name": "x"}]
    with pytest.raises(ValueError):
        model(item, run_type="meta_learn_test_support")
    with pytest.raises(AssertionError):
        model([{"image": torch.zeros(3, 64, 64)}], class_code={"cls_conv": torch.zeros(2, 256), "cls_bias": None},
              run_type="meta_learn_test_instance")
    assert model.device.type == "cuda"


def test_support_output_shapes_like_reference_test(model):
    """tests/code_generator_code_generator_test.py:96-102 of the reference: cls_conv (1, OUT, k, k), cls_bias (1,1,1,1)."""
    from sylph_amd.data import SyntheticSupportSetLoader
    item = next(iter(SyntheticSupportSetLoader(1, 2, 96, 128, seed=1)))
    code = model(item, run_type="meta_learn_test_support")
    assert tuple(code["cls_conv"].shape) == (1, 256, 1, 1) and tuple(code["cls_bias"].shape) == (1, 1, 1, 1)
    codes = [{"support_set_target": torch.tensor(i), "class_name": str(i),
              "class_code": {"cls_conv": torch.rand(1, 256, 1, 1), "cls_bias": torch.rand(1, 1, 1, 1)}} for i in range(3)]
    from sylph_amd.evaluation import inference_normalization
    out = inference_normalization(model, codes)
    assert isinstance(out, list) and len(out) == 3 and tuple(out[0]["class_code"]["cls_bias"].shape) == (1,)


def test_predictor_roundtrip(sd, tmp_path, model):
    from oracle import episode as E
    from sylph_amd.data import SyntheticSupportSetLoader
    from sylph_amd.evaluation import inference_normalization, inference_on_support_set_dataset
    from sylph_amd.predictor import SylphPredictor, resize_image, resize_shortest_edge_shape
    ckpt = str(tmp_path / "model_final.pth")
    torch.save({"model": sd}, ckpt)
    code_dir = str(tmp_path / "codes" / "synthetic_all" / "0")
    sub = inference_on_support_set_dataset(model, SyntheticSupportSetLoader(2, 1, 128, 160, seed=5), output_dir=None)
    sub = inference_normalization(model, sub)
    os.makedirs(code_dir)
    for c in sub:
        c["class_code"] = {k: v.cpu() for k, v in c["class_code"].items()}
        c["class_code"]["cls_conv"] = c["class_code"]["cls_conv"] * 3.0
        torch.save(c, os.path.join(code_dir, f"{c['class_name']}.pth"))
    pred = SylphPredictor("sylph://COCO-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml", ckpt, str(tmp_path / "codes"),
                          test_dataset_names={"all": "synthetic_all"}, dtype="f32")
    pred.min_size, pred.max_size = 96, 160   # keep the test image small
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(90, 130, 3), dtype=np.uint8)
    out = pred._call_few_shot(img, pred.class_codes["all"])["instances"]
    nh, nw = resize_shortest_edge_shape(90, 130, 96, 160)
    x = torch.as_tensor(resize_image(img, nh, nw).astype("float32").transpose(2, 0, 1))
    codes = {k: v.cpu() for k, v in pred.class_codes["all"].items()}
    want = E.forward_instances([x], codes, sd, out_sizes=[(90, 130)])[0]
    assert out.image_size == (90, 130) and len(out) == want["scores"].numel() and len(out) > 0
    np.testing.assert_allclose(out.scores.cpu().numpy(), want["scores"].numpy(), atol=1e-3)
    # boxes are rescaled to the 90 x 130 original by detector_postprocess (factor <= 1 here): same bound
    _assert_boxes(out.pred_boxes.tensor.cpu().numpy(), want["pred_boxes"].numpy(), want["fpn_levels"].numpy())


def test_fused_input_pipeline_is_bit_exact_with_pillow(golden_dir):
    """sylph_preprocess_u8 (resize + BGR + normalise + pad in one kernel) vs outputs of the REAL Pillow (g9_resize.npz):
    in fp32 mode the exported network input equals (PIL pixel - mean) exactly, for every up/down-sampling ratio; the pad
    is zero; RGB input is swapped to BGR."""
    from sylph_amd import synthetic as W
    from sylph_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "g9_resize.npz"))
    eng = Engine(None, dtype="f32")
    eng.load_state_dict(W.backbone_state_dict(0, depth=50))  # the input buffer belongs to the backbone plan
    mean = np.array([103.530, 116.280, 123.675], np.float32).reshape(3, 1, 1)
    cases = g["cases"]
    imgs = [torch.from_numpy(g[f"in{i}"]) for i in range(len(cases))]
    sizes = [(int(c[2]), int(c[3])) for c in cases]
    ph, pw = eng.preprocess_u8(imgs, sizes)  # ONE ragged batch of all cases
    assert ph % 32 == 0 and pw % 32 == 0 and ph >= max(s[0] for s in sizes) and pw >= max(s[1] for s in sizes)
    x = eng.export_input().cpu().numpy()
    for i, (nh, nw) in enumerate(sizes):
        want = g[f"out{i}"].astype(np.float32).transpose(2, 0, 1) - mean
        np.testing.assert_array_equal(x[i, :, :nh, :nw], want)
        assert not x[i, :, nh:, :].any() and not x[i, :, :, nw:].any()
    eng.preprocess_u8([imgs[0].flip(-1)], [sizes[0]], rgb_input=True)  # the same image given as RGB
    np.testing.assert_array_equal(eng.export_input().cpu().numpy()[0, :, :sizes[0][0], :sizes[0][1]],
                                  g["out0"].astype(np.float32).transpose(2, 0, 1) - mean)


def test_predictor_fused_and_host_preprocess_agree(sd, tmp_path, model):
    """SylphPredictor with the fused HIP input pipeline (default) and with the host PIL path give identical detections."""
    from sylph_amd.data import SyntheticSupportSetLoader
    from sylph_amd.evaluation import inference_normalization, inference_on_support_set_dataset
    from sylph_amd.predictor import SylphPredictor
    ckpt = str(tmp_path / "model_final.pth")
    torch.save({"model": sd}, ckpt)
    code_dir = str(tmp_path / "codes" / "synthetic_all" / "0")
    sub = inference_normalization(model, inference_on_support_set_dataset(model, SyntheticSupportSetLoader(2, 1, 128, 160, seed=5)))
    os.makedirs(code_dir)
    for c in sub:
        c["class_code"] = {k: v.cpu() for k, v in c["class_code"].items()}
        c["class_code"]["cls_conv"] = c["class_code"]["cls_conv"] * 3.0
        torch.save(c, os.path.join(code_dir, f"{c['class_name']}.pth"))
    img = np.random.RandomState(1).randint(0, 256, size=(90, 130, 3), dtype=np.uint8)
    outs = []
    for fused in (True, False):
        pred = SylphPredictor("sylph://COCO-Detection/Meta-FCOS/Meta-FCOS-finetune.yaml", ckpt, str(tmp_path / "codes"),
                              test_dataset_names={"all": "synthetic_all"}, dtype="f32", fused_preprocess=fused)
        pred.min_size, pred.max_size = 96, 160
        outs.append(pred._call_few_shot(img, pred.class_codes["all"])["instances"])
    a, b = outs
    assert len(a) == len(b) > 0 and a.image_size == b.image_size == (90, 130)
    assert torch.equal(a.pred_classes, b.pred_classes) and torch.equal(a.scores, b.scores)
    assert torch.equal(a.pred_boxes.tensor, b.pred_boxes.tensor)


def test_msra_pkl_backbone_weights_run_like_reference_keys(tmp_path):
    """An MSRA-style R-50.pkl (Caffe2 blob names, statistics absorbed) read by sylph_amd.checkpoint drives the HIP backbone
    exactly like the same weights given under the reference keys."""
    import pickle
    from sylph_amd import synthetic as W
    from sylph_amd.checkpoint import load_checkpoint_file
    from sylph_amd.engine import Engine
    from test_host_cpu import _to_caffe2
    sd = W.backbone_state_dict(0, depth=50)
    ref = dict(sd)
    for k in list(ref):
        if k.endswith("running_mean"):
            ref[k] = torch.zeros_like(ref[k])
        elif k.endswith("running_var"):
            ref[k] = torch.ones_like(ref[k])
    path = str(tmp_path / "R-50.pkl")
    with open(path, "wb") as f:
        pickle.dump(_to_caffe2(sd), f)
    loaded = load_checkpoint_file(path)
    loaded.update({k: v for k, v in sd.items() if not k.startswith("backbone.bottom_up.")})  # FPN / P6 / P7 come from the detector checkpoint
    imgs = W.synthetic_images(1, 64, 96, seed=2)
    outs = []
    for weights in (ref, loaded):
        eng = Engine(None, dtype="f32")
        eng.load_state_dict(weights)
        eng.preprocess(imgs)
        eng.backbone()
        outs.append([p.cpu() for p in eng.export_pyramid()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_plan_cache_is_bounded_lru(sd, monkeypatch):
    """a stream of distinct padded shapes must not grow HBM without bound.  With SYLPH_MAX_PLANS=3 six shapes
    keep at most three workspaces alive, an evicted shape is rebuilt transparently and gives identical results."""
    from sylph_amd import synthetic as W
    from sylph_amd.engine import Engine
    monkeypatch.setenv("SYLPH_MAX_PLANS", "3")
    eng = Engine(None, dtype="bf16")
    eng.load_state_dict(sd)
    codes = W.synthetic_codes(5, seed=4, scale=3.0)

    def run(h, w):
        eng.preprocess(W.synthetic_images(1, h, w, seed=h + w))
        eng.backbone()
        eng.head(codes["cls_conv"], codes["cls_bias"])
        d = eng.decode()[0]
        return d["scores"].clone(), d["pred_boxes"].clone(), eng.device_bytes()

    first = run(96, 128)
    peak = 0
    for h, w in ((128, 160), (160, 192), (192, 224), (224, 256), (256, 288)):
        peak = max(peak, run(h, w)[2])
    again = run(96, 128)  # evicted by now: rebuilt from scratch
    assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1])
    big3 = 3 * run(256, 288)[2]  # loose upper bound: three times the footprint with the largest shape resident
    assert peak < big3
    monkeypatch.setenv("SYLPH_MAX_PLANS", "64")
    eng2 = Engine(None, dtype="bf16")
    eng2.load_state_dict(sd)
    eng_bytes = []
    for h, w in ((96, 128), (128, 160), (160, 192), (192, 224), (224, 256), (256, 288)):
        eng2.preprocess(W.synthetic_images(1, h, w, seed=h + w)); eng2.backbone()
        eng2.head(codes["cls_conv"], codes["cls_bias"]); eng2.decode()
        eng_bytes.append(eng2.device_bytes())
    assert eng_bytes[-1] > peak, (eng_bytes, peak)  # without eviction the six workspaces add up beyond the bounded engine's peak


def test_result_sink_on_model_outputs(model, sd):
    """Result sink (8f-4) on real device outputs: the batched conversion (one device->host copy) equals the per-image,
    per-field conversion the d2 evaluators do (meta_learn_evaluation.py:428,465), and a single-rank prediction gather
    keeps the rows."""
    from sylph_amd import synthetic as W
    from sylph_amd.data import SyntheticQueryLoader
    from sylph_amd.evaluation import detection_rows_to_coco, detections_to_coco_rows, detections_to_tensor, gather_detection_rows
    codes = W.synthetic_codes(3, seed=4, scale=3.0)
    batch = next(iter(SyntheticQueryLoader(3, 120, 152, batch_size=3, seed=8)))
    outs = model(batch, class_code={k: v.cuda() for k, v in codes.items()}, run_type="meta_learn_test_instance")
    ids = [b["image_id"] + 100 for b in batch]
    rows = detections_to_coco_rows(outs, ids, {0: 7, 1: 8, 2: 9})
    want = []
    for img_id, o in zip(ids, outs):  # the reference way: per image, per field
        i = o["instances"]
        bx, sc, cl = i.pred_boxes.tensor.cpu().tolist(), i.scores.cpu().tolist(), i.pred_classes.cpu().tolist()
        for b, s, c in zip(bx, sc, cl):
            want.append({"image_id": img_id, "category_id": {0: 7, 1: 8, 2: 9}[c], "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]], "score": s})
    assert len(rows) == len(want) > 0
    for r, w in zip(rows, want):
        assert r["image_id"] == w["image_id"] and r["category_id"] == w["category_id"]
        np.testing.assert_allclose(r["bbox"], w["bbox"], rtol=1e-6, atol=1e-4)
        assert abs(r["score"] - w["score"]) < 1e-6
    t = detections_to_tensor(outs, ids)
    assert t.is_cuda and t.shape == (len(want), 8)
    g = gather_detection_rows(t, capacity=len(want) + 5)
    assert len(detection_rows_to_coco(g)) == len(want)


def test_eval_with_pretrained_code_uses_the_checkpoints_cls_logits(model, sd):
    """class_code=None (meta_learn_evaluation.py:376-378 eval_with_pretrained_code) -> MetaFCOSHead.forward_base_train:
    logits = cls_logits(cls_tower) with the checkpoint's own 1x1 classifier.  Must equal the class-conditional path fed with
    those weights as codes, and a plain fp32 conv of the oracle's cls-tower output."""
    from oracle import backbone as OB, head as OH
    from sylph_amd import synthetic as W
    from sylph_amd.evaluation import inference_on_dataset_with_class_codes
    w = sd["proposal_generator.fcos_head.cls_logits.weight"]
    b = sd["proposal_generator.fcos_head.cls_logits.bias"]
    assert w.shape[2:] == (1, 1)
    imgs = W.synthetic_images(2, 96, 128, seed=21)
    batch = [{"image": im, "height": 96, "width": 128, "image_id": i} for i, im in enumerate(imgs)]
    a = model(batch, class_code=None, run_type="meta_learn_test_instance")
    e = model(batch, class_code={"cls_conv": w.cuda(), "cls_bias": b.cuda()}, run_type="meta_learn_test_instance")
    for x, y in zip(a, e):
        assert torch.equal(x["instances"].pred_boxes.tensor, y["instances"].pred_boxes.tensor)
        assert torch.equal(x["instances"].scores, y["instances"].scores)
        assert torch.equal(x["instances"].pred_classes, y["instances"].pred_classes)
    # the logits of the pretrained path against the oracle's tower + an fp32 conv2d with the same weights
    got = model.engine.export_head()[0]
    x0, _ = OB.preprocess(imgs)
    feats = OB.backbone_fpn(x0, sd, 50)
    for l in range(5):
        t = OH.tower(feats[l], sd, "proposal_generator.fcos_head.cls_tower")
        ref = torch.nn.functional.conv2d(t, w, b)
        assert (got[l].cpu() - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
    col = _Collect()
    res = inference_on_dataset_with_class_codes(model, [batch], col, None, eval_with_pretrained_code=True)
    assert res == {"n": 2}
    with pytest.raises(AssertionError):
        inference_on_dataset_with_class_codes(model, [batch], col, {"cls_conv": w}, eval_with_pretrained_code=True)


def test_base_detector_inference_run_type_none(sd):
    """run_type=None on a NON-episodic model = MetaProposalNetwork.forward_base_detector (meta_one_stage_detector.py:298-323,
    436-441): plain FCOS with the checkpoint's cls_logits; no code generator is built or needed.  Must give the detections of
    the episodic model fed with the same weights as class codes."""
    from sylph_amd import synthetic as W
    from sylph_amd.runner import MetaFCOSRunner
    r = MetaFCOSRunner()
    cfg = r.get_default_cfg()
    cfg.MODEL.META_LEARN.EPISODIC_LEARNING = False
    cfg.MODEL.FCOS.CLS_LOGITS_KERNEL_SIZE = 1
    cfg.MODEL.FCOS.NUM_CLASSES = int(sd["proposal_generator.fcos_head.cls_logits.weight"].shape[0])
    base = r.build_model(cfg, dtype="f32")
    assert base.code_generator is None
    base.load_state_dict({k: v for k, v in sd.items() if not k.startswith("code_generator.")})
    base.eval()
    imgs = W.synthetic_images(2, 96, 128, seed=22)
    batch = [{"image": im, "height": 96, "width": 128} for im in imgs]
    got = base(batch)
    runner, ecfg = _cfg()
    epi = runner.build_model(ecfg, dtype="f32")
    epi.load_state_dict(sd)
    epi.eval()
    w, b = sd["proposal_generator.fcos_head.cls_logits.weight"], sd["proposal_generator.fcos_head.cls_logits.bias"]
    exp = epi(batch, class_code={"cls_conv": w.cuda(), "cls_bias": b.cuda()}, run_type="meta_learn_test_instance")
    assert len(got) == 2
    for x, y in zip(got, exp):
        assert torch.equal(x["instances"].pred_boxes.tensor, y["instances"].pred_boxes.tensor)
        assert torch.equal(x["instances"].scores, y["instances"].scores)
    with pytest.raises(AssertionError):
        base(batch, class_code={"cls_conv": w.cuda(), "cls_bias": b.cuda()}, run_type="meta_learn_test_instance")
    with pytest.raises(NotImplementedError):
        epi(batch)  # episodic models refuse run_type=None like the reference


def test_repeated_steps_are_bit_identical(sd):
    """Race check of the hand-synchronised kernels (counted vmcnt / lgkmcnt waits, raw barriers, LDS-DMA double buffers): the same
    bf16 batch through preprocess -> backbone -> head -> decode must give bit-identical pyramids, head outputs and detections
    every time (every reduction runs in a fixed order).  tools/soak_determinism.py is the long version."""
    from sylph_amd import synthetic as W
    from sylph_amd.engine import Engine
    _, cfg = _cfg()
    eng = Engine(cfg, dtype="bf16")
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    w = (torch.randn(5, 256, 1, 1, generator=g) * 0.05).cuda()
    b = torch.full((5,), -2.0).cuda()
    imgs = [im.cuda() for im in W.synthetic_images(3, 352, 480, seed=31)]
    ref = None
    for it in range(6):
        eng.preprocess(imgs); eng.backbone(); eng.head(w, b)
        lo, rg, ct, _ = eng.export_head()
        dets = eng.decode()
        cur = [t.clone() for t in eng.export_pyramid()] + [t.clone() for t in lo + rg + ct] + \
              [d["pred_boxes"].clone() for d in dets] + [d["scores"].clone() for d in dets]
        if ref is None:
            ref = cur
            continue
        for k, (x, y) in enumerate(zip(ref, cur)):
            assert x.shape == y.shape and torch.equal(x, y), f"iteration {it}: tensor {k} differs"


def test_registry_hosts_a_different_component(sd):
    """The registries are real plugin points (sylph/modeling/code_generator/build.py:18-39): a code generator registered under
    another name is built from the yaml key, bound to the model's HIP context and called by forward_class_code."""
    from sylph_amd import modeling as M
    from sylph_amd import synthetic as W
    from sylph_amd.structures import Boxes, Instances
    calls = []

    @M.CODE_GENERATOR_REGISTRY.register()
    class HalvedCodeGenerator(M.CodeGenerator):
        def __call__(self, boxes=None, **kw):
            out = super().__call__(boxes, **kw)
            if isinstance(out, dict):
                calls.append(int(boxes.shape[0]))
                out["cls_conv"] = out["cls_conv"] * 0.5
            return out

    runner, cfg = _cfg()
    cfg.MODEL.META_LEARN.CODE_GENERATOR.NAME = "HalvedCodeGenerator"
    m = runner.build_model(cfg, dtype="f32")
    assert isinstance(m.code_generator, HalvedCodeGenerator) and m.code_generator.engine is m.engine
    m.load_state_dict(sd)
    m.eval()
    cfg2 = cfg.clone()
    cfg2.MODEL.META_LEARN.CODE_GENERATOR.NAME = "CodeGenerator"
    ref = runner.build_model(cfg2, dtype="f32")
    ref.load_state_dict(sd)
    ref.eval()
    imgs = W.synthetic_images(2, 96, 128, seed=41)
    sup = []
    for im in imgs:
        inst = Instances((96, 128))
        inst.gt_boxes = Boxes(torch.tensor([[10.0, 12.0, 70.0, 80.0]]))
        inst.gt_classes = torch.zeros(1, dtype=torch.long)
        sup.append({"image": im, "instances": inst})
    item = [{"support_set": sup, "support_set_target": torch.tensor(0), "class_name": "x"}]
    a = m(item, run_type="meta_learn_test_support")
    b = ref(item, run_type="meta_learn_test_support")
    assert calls == [2]
    assert torch.allclose(a["cls_conv"], b["cls_conv"] * 0.5, rtol=0, atol=0) and torch.equal(a["cls_bias"], b["cls_bias"])
    with pytest.raises(KeyError):
        M.CODE_GENERATOR_REGISTRY.get("NoSuchGenerator")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_support_classes_batched_equal_one_class_per_call(sd, dtype):
    """Support-path batching (sylph_codegen_classes): 3 classes x 2 shots through ONE backbone / code-generator batch against the
    reference protocol of one class per call (meta_one_stage_detector.py:229-254).  fp32: bit-identical (per-image arithmetic does
    not depend on the batch around it); bf16: to bf16 rounding of the stored activations (kernel selection follows the launch
    size).  Then the evaluation loop: grouped and un-grouped support inference return the same records."""
    from sylph_amd.data import SyntheticSupportSetLoader
    from sylph_amd.evaluation import inference_on_support_set_dataset
    runner, cfg = _cfg()
    m = runner.build_model(cfg, dtype=dtype)
    m.load_state_dict(sd)
    m.eval()
    items = list(SyntheticSupportSetLoader(3, 2, 128, 160, seed=9))
    one = [m(it, run_type="meta_learn_test_support") for it in items]
    one = [{k: v.clone() for k, v in c.items()} for c in one]
    many = m.forward_class_codes(items)
    assert len(many) == 3
    for a, b in zip(one, many):
        assert tuple(b["cls_conv"].shape) == (1, 256, 1, 1) and tuple(b["cls_bias"].shape) == (1, 1, 1, 1)
        if dtype == "f32":
            assert torch.equal(a["cls_conv"], b["cls_conv"]) and torch.equal(a["cls_bias"], b["cls_bias"])
        else:
            scale = float(a["cls_conv"].abs().max())
            assert float((a["cls_conv"] - b["cls_conv"]).abs().max()) <= 2e-2 * scale
    grouped = inference_on_support_set_dataset(m, SyntheticSupportSetLoader(3, 2, 128, 160, seed=9))
    assert [int(r["support_set_target"]) for r in grouped] == [0, 1, 2] and [r["class_name"] for r in grouped] == [it[0]["class_name"] for it in items]
    for r, b in zip(grouped, many):
        assert torch.equal(r["class_code"]["cls_conv"], b["cls_conv"].cpu())


def _mixed_size_support_items(sizes_per_class, seed=5):
    """Loader items (one class each) whose support images have the given (h, w) sizes; one box per image."""
    from sylph_amd import synthetic as W
    from sylph_amd.structures import Boxes, Instances
    items = []
    for c, sizes in enumerate(sizes_per_class):
        recs = []
        for s, (h, w) in enumerate(sizes):
            im = W.synthetic_images(1, h, w, seed=seed * 100 + c * 10 + s)[0]
            inst = Instances((h, w))
            inst.gt_boxes = Boxes(torch.tensor([[8.0 + c, 6.0 + s, 0.7 * w, 0.8 * h]]))
            inst.gt_classes = torch.tensor([c])
            recs.append({"image": im, "instances": inst, "height": h, "width": w})
        items.append([{"support_set": recs, "support_set_target": torch.tensor(c), "class_name": f"class_{c}"}])
    return items


def test_support_batching_respects_per_class_padding(sd):
    """with mixed image sizes a class may only share a batch with classes that are padded to the same (H, W) when run
    alone (the reference pads ONE class per call, meta_one_stage_detector.py:229-254; the padded size decides the border
    activations).  Classes of kind A pad to 128x160 (the FIRST image of the first one is 90x120, exactly the first image of a
    kind-B class that pads to 96x128: the old first-image guard grouped those two).  fp32 codes must be bit-identical to one class per call, both
    through forward_class_codes (falls back) and through the evaluation loop (groups only 2 + 3)."""
    from sylph_amd.evaluation import class_padded_hw, inference_on_support_set_dataset
    runner, cfg = _cfg()
    m = runner.build_model(cfg, dtype="f32")
    m.load_state_dict(sd)
    m.eval()
    A = [[(90, 120), (128, 150)], [(120, 160), (128, 130)], [(128, 160), (100, 100)]]  # pad to 128x160 (first image small / ragged)
    Bs = [[(90, 120), (96, 128)], [(96, 100), (70, 128)]]  # pad to 96x128
    sizes = [A[0], Bs[0], A[1], A[2], Bs[1], Bs[0], A[0], A[2]]
    items = _mixed_size_support_items(sizes)
    pads = [class_padded_hw(it[0]["support_set"]) for it in items]
    assert pads == [(128, 160), (96, 128), (128, 160), (128, 160), (96, 128), (96, 128), (128, 160), (128, 160)]
    one = [{k: v.clone() for k, v in m(it, run_type="meta_learn_test_support").items()} for it in items]
    calls = []
    orig = m.forward_class_codes

    def spy(group):
        calls.append(len(group))
        return orig(group)

    m.forward_class_codes = spy
    many = orig(items)  # mixed padded sizes: must fall back to one call per class
    for a, b in zip(one, many):
        assert torch.equal(a["cls_conv"], b["cls_conv"]) and torch.equal(a["cls_bias"], b["cls_bias"])
    grouped = inference_on_support_set_dataset(m, items)
    # the loop flushes at every padded-size change (and once at the end of its 5 warm-up items): only (2, 3) and (6, 7) share a batch
    assert calls == [2, 2], calls
    for a, r in zip(one, grouped):
        assert torch.equal(a["cls_conv"].cpu(), r["class_code"]["cls_conv"]) and torch.equal(a["cls_bias"].cpu(), r["class_code"]["cls_bias"])
    same = orig([items[2], items[3], items[6]])  # equal padded sizes, ragged images inside: one shared batch, identical codes
    for a, b in zip([one[2], one[3], one[6]], same):
        assert torch.equal(a["cls_conv"], b["cls_conv"]) and torch.equal(a["cls_bias"], b["cls_bias"])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_base_detector_with_3x3_cls_logits(sd, dtype):
    """MODEL.FCOS.CLS_LOGITS_KERNEL_SIZE = 3 (the adet-style classifier, fcos.py:418-427; Sylph's default_configs.py:48 switches it to 1):
    run_type None on a non-episodic model runs `cls_logits(cls_tower)` as a real 3x3 conv (sylph_fcos_head_pretrained).  fp32: logits
    <= 1e-3 of the oracle's tower + F.conv2d, detections = the oracle decoder on those head outputs; bf16: finite + close logits."""
    from oracle import backbone as OB, decode as OD, head as OH
    from sylph_amd import synthetic as W
    from sylph_amd.runner import MetaFCOSRunner
    g = torch.Generator().manual_seed(77)
    sd3 = {k: v for k, v in sd.items() if not k.startswith("code_generator.")}
    w3 = torch.randn(7, 256, 3, 3, generator=g) * (1.0 / (9 * 256)) ** 0.5 * 2.0
    b3 = torch.full((7,), -2.0) + 0.2 * torch.randn(7, generator=g)
    sd3["proposal_generator.fcos_head.cls_logits.weight"], sd3["proposal_generator.fcos_head.cls_logits.bias"] = w3, b3
    r = MetaFCOSRunner()
    cfg = r.get_default_cfg()
    cfg.MODEL.META_LEARN.EPISODIC_LEARNING = False
    cfg.MODEL.FCOS.CLS_LOGITS_KERNEL_SIZE = 3
    cfg.MODEL.FCOS.NUM_CLASSES = 7
    base = r.build_model(cfg, dtype=dtype)
    base.load_state_dict(sd3)
    base.eval()
    imgs = W.synthetic_images(2, 96, 128, seed=23)
    batch = [{"image": im, "height": 96, "width": 128} for im in imgs]
    got = base(batch)
    lo, rg, ct, io = [[t.cpu() for t in ts] for ts in base.engine.export_head()]
    x0, sizes = OB.preprocess(imgs)
    feats = OB.backbone_fpn(x0, sd3, 50)
    tol = 1e-3 if dtype == "f32" else 6e-2
    for l in range(5):
        t = OH.tower(feats[l], sd3, "proposal_generator.fcos_head.cls_tower")
        ref = torch.nn.functional.conv2d(t, w3, b3, padding=1)
        assert tuple(lo[l].shape) == tuple(ref.shape)
        assert (lo[l] - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), l
    if dtype == "f32":
        want = OD.predict_proposals(lo, rg, ct, io)
        for i in range(2):
            wv = OD.detector_postprocess(want[i], sizes[i], 96, 128)
            inst = got[i]["instances"]
            assert len(inst) == wv["scores"].numel() > 0
            np.testing.assert_array_equal(inst.pred_classes.cpu().numpy(), wv["pred_classes"].numpy())
            np.testing.assert_array_equal(inst.locations.cpu().numpy(), wv["locations"].numpy())
            np.testing.assert_allclose(inst.scores.cpu().numpy(), wv["scores"].numpy(), atol=1e-5)
    else:
        assert all(torch.isfinite(o["instances"].scores).all() for o in got)
