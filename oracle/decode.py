"""Oracle: FCOS locations, per-level decode, class-aware NMS, top-k keep, postprocess.
TEST INFRASTRUCTURE.  fp32 torch/numpy CPU.

Sylph-owned arithmetic followed (paths relative to /root/reference):
  * sylph/modeling/meta_fcos/fcos.py:270-282                 compute_locations (formula in the
    docstring :272; body is AdelaiDet adet/utils/comm.py compute_locations, restated)
  * sylph/modeling/meta_fcos/fcos_outputs.py:743-812         predict_proposals
  * sylph/modeling/meta_fcos/fcos_outputs.py:904-1008        forward_for_single_feature_map
  * sylph/modeling/meta_fcos/fcos_outputs.py:1010-1028       select_over_all_levels
  * sylph/modeling/meta_arch/meta_one_stage_detector.py:288-296  postprocess loop
Third-party arithmetic restated (not under /root/reference; parity unpinned by the reference):
  * adet.layers.ml_nms -> detectron2.layers.batched_nms -> torchvision.ops.nms: greedy NMS per
    class, boxes visited in descending score order, suppress when IoU > thresh
    (IoU = inter / (area_i + area_j - inter), no +1), result sorted by descending score.
    Restated as the exact per-class form (torchvision ``_batched_nms_vanilla``); the
    coordinate-offset variant differs only by fp32 rounding of the offset boxes.
  * adet detector_postprocess -> detectron2 detector_postprocess: scale boxes by
    (out_w / img_w, out_h / img_h), clip to the output size, drop empty boxes.

Determinism contract (shared with the HIP path; the reference leaves these unspecified):
  * candidates are ordered (level, location, class) -- the ``nonzero`` order of
    fcos_outputs.py:967-969 concatenated over levels (:808-809);
  * pre-NMS top-k (``topk(sorted=False)``, :980-982) keeps the k largest ``cls*ctr`` values, ties
    at the k-th value resolved towards the lower (location, class) index, order preserved;
  * NMS visits boxes by descending ``scores`` (= sqrt(cls*ctr)), ties by ascending candidate index.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


def compute_locations(h: int, w: int, stride: int) -> torch.Tensor:
    """(h*w, 2) fp32, x = j*s + s//2, y = i*s + s//2, row-major over (i, j).  fcos.py:270-282."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + stride // 2


def decode_level(locations: torch.Tensor, logits: torch.Tensor, reg: torch.Tensor, ctr: torch.Tensor,
                 iou: torch.Tensor, pre_nms_thresh: float = 0.05, pre_nms_topk: int = 1000,
                 thresh_with_ctr: bool = False, box_quality: Sequence[str] = ("ctrness",), owd: bool = False) -> List[Dict]:
    """fcos_outputs.py:904-1008.  ``reg`` is already multiplied by the level stride (:786).  owd: MODEL.PROPOSAL_GENERATOR.OWD
    (:913-916): the class probabilities are replaced by ONE all-ones class, and -- :937 ``thresh_with_ctr or OWD`` / :951 ``not
    thresh_with_ctr and not OWD`` -- the box quality is multiplied in BEFORE the ``> pre_nms_thresh`` test, i.e. the OWD candidates
    are the locations whose quality alone clears the threshold (round 5: rounds <= 4 thresholded the constant 1)."""
    N, C, H, W = logits.shape
    p = logits.permute(0, 2, 3, 1).reshape(N, -1, C).sigmoid()
    if owd:
        p = torch.ones_like(p)[:, :, [0]]
        C = 1
    box_reg = reg.view(N, 4, H, W).permute(0, 2, 3, 1).reshape(N, -1, 4)
    c = ctr.view(N, 1, H, W).permute(0, 2, 3, 1).reshape(N, -1).sigmoid()
    q = iou.view(N, 1, H, W).permute(0, 2, 3, 1).reshape(N, -1).sigmoid()
    bq = sorted(box_quality)
    if bq == ["ctrness"]:
        quality = c[:, :, None]
    elif bq == ["iou"]:
        quality = q[:, :, None]
    elif bq == ["ctrness", "iou"]:
        quality = torch.sqrt(q[:, :, None] * c[:, :, None])
    else:
        raise NotImplementedError()
    quality_first = thresh_with_ctr or owd                      # :937
    if quality_first:
        p = p * quality
    cand = p > pre_nms_thresh
    top_n = cand.reshape(N, -1).sum(1).clamp(max=pre_nms_topk)
    if not quality_first:                                       # :951
        p = p * quality
    results = []
    for i in range(N):
        vals = p[i][cand[i]]
        nz = cand[i].nonzero()
        loc_idx, cls_idx = nz[:, 0], nz[:, 1]
        k = int(top_n[i])
        if vals.numel() > k:
            # deterministic top-k: k largest, ties -> lower index, original order kept
            order = np.lexsort((np.arange(vals.numel()), -vals.numpy().astype(np.float64)))
            sel = torch.from_numpy(np.sort(order[:k]))
            vals, loc_idx, cls_idx = vals[sel], loc_idx[sel], cls_idx[sel]
        r = box_reg[i][loc_idx]
        l = locations[loc_idx]
        boxes = torch.stack([l[:, 0] - r[:, 0], l[:, 1] - r[:, 1], l[:, 0] + r[:, 2], l[:, 1] + r[:, 3]], dim=1)
        results.append({"pred_boxes": boxes, "scores": torch.sqrt(vals), "pred_classes": cls_idx,
                        "locations": l, "loc_index": loc_idx})
    return results


def nms_per_class(boxes: np.ndarray, scores: np.ndarray, classes: np.ndarray, thresh: float) -> np.ndarray:
    """Greedy class-aware NMS (torchvision nms arithmetic, fp32).  Returns kept indices in
    descending-score order (ties: ascending index)."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    boxes = boxes.astype(np.float32)
    order = np.lexsort((np.arange(n), -scores.astype(np.float64)))
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = ((x2 - x1) * (y2 - y1)).astype(np.float32)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for oi in range(n):
        i = order[oi]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        rest = rest[(~suppressed[rest]) & (classes[rest] == classes[i])]
        if rest.size == 0:
            continue
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), (xx2 - xx1).astype(np.float32))
        h = np.maximum(np.float32(0), (yy2 - yy1).astype(np.float32))
        inter = (w * h).astype(np.float32)
        union = ((areas[i] + areas[rest]).astype(np.float32) - inter).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = (inter / union).astype(np.float32)
        suppressed[rest[ovr > np.float32(thresh)]] = True
    return np.asarray(keep, dtype=np.int64)


def select_over_all_levels(inst: Dict, nms_thresh: float = 0.6, post_nms_topk: int = 100) -> Dict:
    """fcos_outputs.py:1010-1028: ml_nms, then keep scores >= k-th largest (ties kept)."""
    if nms_thresh > 0:
        keep = nms_per_class(inst["pred_boxes"].numpy(), inst["scores"].numpy(),
                             inst["pred_classes"].numpy(), nms_thresh)
        keep = torch.from_numpy(keep)
        inst = {k: v[keep] for k, v in inst.items()}
        inst["nms_keep"] = keep
    n = inst["scores"].numel()
    if n > post_nms_topk > 0:
        thr, _ = torch.kthvalue(inst["scores"], n - post_nms_topk + 1)
        sel = torch.nonzero(inst["scores"] >= thr.item()).squeeze(1)
        inst = {k: v[sel] for k, v in inst.items()}
    return inst


def predict_proposals(logits, regs, ctrs, ious, strides=(8, 16, 32, 64, 128), pre_nms_thresh=0.05,
                      pre_nms_topk=1000, nms_thresh=0.6, post_nms_topk=100, thresh_with_ctr=False,
                      box_quality=("ctrness",), owd=False) -> List[Dict]:
    """fcos_outputs.py:743-812.  Per image dict: pred_boxes, scores, pred_classes, locations,
    fpn_levels (+ loc_index, cand_index = index into the concatenated pre-NMS candidate list)."""
    per_level = []
    for level, (o, r, c, q) in enumerate(zip(logits, regs, ctrs, ious)):
        h, w = o.shape[-2:]
        loc = compute_locations(h, w, strides[level])
        res = decode_level(loc, o, r * strides[level], c, q, pre_nms_thresh, pre_nms_topk,
                           thresh_with_ctr, box_quality, owd)
        for d in res:
            d["fpn_levels"] = torch.full((d["scores"].numel(),), level, dtype=torch.long)
        per_level.append(res)
    out = []
    for i in range(logits[0].shape[0]):
        cat = {k: torch.cat([lv[i][k] for lv in per_level], dim=0) for k in per_level[0][i].keys()}
        cat["cand_index"] = torch.arange(cat["scores"].numel())
        out.append(select_over_all_levels(cat, nms_thresh, post_nms_topk))
    return out


def detector_postprocess(inst: Dict, image_size: Tuple[int, int], out_h: int, out_w: int) -> Dict:
    """detectron2/adet detector_postprocess restated; call site meta_one_stage_detector.py:292-295."""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b = inst["pred_boxes"].clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0].clamp_(min=0, max=out_w)
    b[:, 1].clamp_(min=0, max=out_h)
    b[:, 2].clamp_(min=0, max=out_w)
    b[:, 3].clamp_(min=0, max=out_h)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    res = {k: v[keep] for k, v in inst.items()}
    res["pred_boxes"] = b[keep]
    return res


def explain_absence(heads, i: int, key: Tuple[int, int, int], other_final_keys, eps_val: float, eps_iou: float, rel: bool = False,
                    strides=(8, 16, 32, 64, 128), pre_nms_thresh=0.05, pre_nms_topk=1000, nms_thresh=0.6, post_nms_topk=100,
                    thresh_with_ctr=False, box_quality=("ctrness",)) -> Tuple[str, float, bool]:
    """Why is the candidate ``key`` = (level, location index, class) NOT among the detections that ``heads`` (per-level logits / reg /
    ctrness / iou of the whole batch) yield for image ``i``?  Walks the decision chain of fcos_outputs.py:904-1028 on those head
    outputs and returns (reason, margin, marginal): the FIRST decision that removes the candidate, its distance to that decision's
    boundary, and whether the distance is within tolerance -- i.e. whether a perturbation of the head outputs of that size (the other
    pipeline's numerical noise) can flip the decision:

      "threshold"   sigmoid(logit) (x quality with THRESH_WITH_CTR) <= pre_nms_thresh            margin = thresh - p
      "level_topk"  cls x quality is below the k-th largest of its (image, level)               margin = kth - val
      "post_topk"   survives NMS, but its score is below the k-th largest kept score             margin = kth^2 - val (val = score^2)
      "nms"         suppressed: EVERY suppressor s (kept, same class, ahead in the order, IoU > nms_thresh) must be escapable --
                    IoU - nms_thresh <= eps_iou, or the order can flip (|val - val_s| within tolerance), or s is itself absent
                    from the other pipeline's detections (``other_final_keys``; s then gets its own explanation)   margin = the worst of them
      "present"     the candidate IS a detection of ``heads`` (nothing to explain)

    Tolerances: |margin| <= eps_val (absolute on cls x quality, or relative to the boundary value with rel=True), eps_iou on IoU.
    Test infrastructure for the end-to-end parity statements."""
    lvl, loc, cls = key
    logits, regs, ctrs, ious = heads
    tol = (lambda b: eps_val * max(abs(b), 1e-12)) if rel else (lambda b: eps_val)
    per_level = []
    for level, (o, r, c, q) in enumerate(zip(logits, regs, ctrs, ious)):
        h, w = o.shape[-2:]
        res = decode_level(compute_locations(h, w, strides[level]), o[i:i + 1], r[i:i + 1] * strides[level], c[i:i + 1], q[i:i + 1],
                           pre_nms_thresh, pre_nms_topk, thresh_with_ctr, box_quality)[0]
        res["fpn_levels"] = torch.full((res["scores"].numel(),), level, dtype=torch.long)
        per_level.append(res)
    # (1) threshold
    o = logits[lvl][i]
    C, H, W = o.shape
    p = torch.sigmoid(o[cls].reshape(-1)[loc]).item()
    cq = torch.sigmoid(ctrs[lvl][i].reshape(-1)[loc]).item()
    iq = torch.sigmoid(ious[lvl][i].reshape(-1)[loc]).item()
    bq = sorted(box_quality)
    quality = cq if bq == ["ctrness"] else (iq if bq == ["iou"] else (iq * cq) ** 0.5)
    p_thr = p * quality if thresh_with_ctr else p
    val = p * quality
    if not p_thr > pre_nms_thresh:
        m = pre_nms_thresh - p_thr
        return "threshold", m, m <= tol(pre_nms_thresh)
    # (2) per-level top-k
    lv = per_level[lvl]
    mine = ((lv["loc_index"] == loc) & (lv["pred_classes"] == cls)).nonzero().reshape(-1)
    if mine.numel() == 0:
        kth = float((lv["scores"] ** 2).min())
        m = kth - val
        return "level_topk", m, m <= tol(kth)
    # (3) NMS / (4) post-NMS top-k on the concatenated candidates
    cat = {k: torch.cat([d[k] for d in per_level], dim=0) for k in per_level[0].keys()}
    me = int(sum(d["scores"].numel() for d in per_level[:lvl]) + mine[0])
    keep = nms_per_class(cat["pred_boxes"].numpy(), cat["scores"].numpy(), cat["pred_classes"].numpy(), nms_thresh)
    kept_scores = cat["scores"][torch.from_numpy(keep)]
    if me in set(keep.tolist()):
        n = kept_scores.numel()
        if n > post_nms_topk > 0:
            thr, _ = torch.kthvalue(kept_scores, n - post_nms_topk + 1)
            if float(cat["scores"][me]) < float(thr):
                m = float(thr) ** 2 - val
                return "post_topk", m, m <= tol(float(thr) ** 2)
        return "present", 0.0, True
    b = cat["pred_boxes"].numpy().astype(np.float32)
    sc = cat["scores"].numpy()
    worst, ok_all = 0.0, True
    found = False
    for s in keep.tolist():
        if int(cat["pred_classes"][s]) != cls or not (sc[s] > sc[me] or (sc[s] == sc[me] and s < me)):
            continue
        iw = max(np.float32(0), min(b[s, 2], b[me, 2]) - max(b[s, 0], b[me, 0]))
        ih = max(np.float32(0), min(b[s, 3], b[me, 3]) - max(b[s, 1], b[me, 1]))
        inter = np.float32(iw * ih)
        union = np.float32((b[s, 2] - b[s, 0]) * (b[s, 3] - b[s, 1]) + (b[me, 2] - b[me, 0]) * (b[me, 3] - b[me, 1]) - inter)
        ov = float(inter / union) if union > 0 else 0.0
        if ov > nms_thresh:
            found = True
            skey = (int(cat["fpn_levels"][s]), int(cat["loc_index"][s]), int(cat["pred_classes"][s]))
            d_iou = ov - nms_thresh
            d_ord = abs(float(sc[s]) ** 2 - val)
            esc = d_iou <= eps_iou or d_ord <= tol(val) or (other_final_keys is not None and skey not in other_final_keys)
            ok_all = ok_all and esc
            worst = max(worst, min(d_iou, d_ord))
    if not found:  # suppressed by a box that was itself cut afterwards cannot happen in greedy NMS; report it as unexplained
        return "nms", float("inf"), False
    return "nms", worst, ok_all
