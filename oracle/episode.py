"""Oracle: class-code bookkeeping and the full episodic inference pipeline.  TEST INFRASTRUCTURE.

Follows (paths relative to /root/reference):
  * sylph/evaluation/meta_learn_evaluation.py:71-103   format_class_codes_shared
  * sylph/evaluation/meta_learn_evaluation.py:176-188  base-class weighted accumulation
  * sylph/modeling/code_generator/utils.py:357-427     convert_list_to_dict / reduce_class_code
  * sylph/runner/meta_fcos_runner.py:381-439           _gather_class_code (rank-order flatten)
  * sylph/modeling/meta_arch/meta_one_stage_detector.py:229-296  forward_class_code / forward_instances
"""
from collections import OrderedDict, defaultdict
from typing import Any, Dict, List, Sequence

import torch

from . import backbone as _bb
from . import codegen as _cg
from . import decode as _dec
from . import head as _head


def format_class_codes_shared(class_codes: List[Dict[str, Any]]) -> Dict[str, torch.Tensor]:
    """meta_learn_evaluation.py:71-103: order by support_set_target, cat, flatten cls_bias."""
    n = len(class_codes)
    if n == 0:
        return class_codes
    outs = defaultdict(list)
    for k in class_codes[0]["class_code"].keys():
        outs[k] = [None] * n
    for code in class_codes:
        for k, v in code["class_code"].items():
            if k == "snnl":
                continue
            outs[k][int(code["support_set_target"])] = v
    final = {}
    for k, v in outs.items():
        final[k] = torch.cat(v, dim=0)
        if k == "cls_bias":
            final[k] = final[k].view(final[k].numel())
    return final


def reduce_class_code(out_codes: List[Dict]) -> List[Dict]:
    """code_generator/utils.py:397-427: per class id sum the chunk codes (already weighted by
    len/total_len), divide by acc_weight when |1 - acc| > 1e-6, drop acc_weight."""
    if len(out_codes) == 0:
        return out_codes
    keys = list(out_codes[0]["class_code"].keys())
    by_cid, other = OrderedDict(), {}
    for c in out_codes:
        cid = int(c["support_set_target"])
        by_cid.setdefault(cid, []).append(c["class_code"])
        if cid not in other:
            other[cid] = {k: v for k, v in c.items() if k != "class_code"}
    results = []
    for cid, lst in by_cid.items():
        r = dict(other[cid])
        cc = {}
        for k in keys:
            acc = 0
            for item in lst:
                acc = acc + item[k]
            cc[k] = acc
        aw = float(cc["acc_weight"])
        if abs(1.0 - aw) > 1e-6:
            cc["cls_conv"] = cc["cls_conv"] / aw
            cc["cls_bias"] = cc["cls_bias"] / aw
            if "cls_weight_norm" in cc:
                cc["cls_weight_norm"] = cc["cls_weight_norm"] / aw
        del cc["acc_weight"]
        r["class_code"] = cc
        results.append(r)
    return results


def gather_class_code(per_rank_codes: Sequence[List[Dict]], reduce: bool = False) -> List[Dict]:
    """meta_fcos_runner.py:381-439: flatten the per-rank lists in rank order, optional reduce."""
    out = [c for sub in per_rank_codes for c in sub]
    return reduce_class_code(out) if reduce else out


def forward_class_code(support_images: List[torch.Tensor], boxes: torch.Tensor, sd, depth: int = 50, **kw):
    """meta_one_stage_detector.py:229-254: S support images of ONE class, one gt box each."""
    x, _ = _bb.preprocess(support_images)
    feats = _bb.backbone_fpn(x, sd, depth)
    return _cg.code_generator(feats, boxes, sd, **kw)


def forward_instances(images: List[torch.Tensor], class_codes: Dict[str, torch.Tensor], sd, depth: int = 50,
                      out_sizes=None, post_nms_topk: int = 100, **decode_kw) -> List[Dict]:
    """meta_one_stage_detector.py:261-296 (eval, no gt)."""
    x, sizes = _bb.preprocess(images)
    feats = _bb.backbone_fpn(x, sd, depth)
    logits, regs, ctrs, ious = _head.fcos_head(feats, sd, class_codes)
    props = _dec.predict_proposals(logits, regs, ctrs, ious, post_nms_topk=post_nms_topk, **decode_kw)
    res = []
    for i, p in enumerate(props):
        oh, ow = out_sizes[i] if out_sizes is not None else sizes[i]
        res.append(_dec.detector_postprocess(p, sizes[i], oh, ow))
    return res
