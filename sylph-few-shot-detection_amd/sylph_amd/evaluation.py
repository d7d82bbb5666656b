"""Episodic inference loops and class-code formatting with the reference's signatures
(sylph/evaluation/meta_learn_evaluation.py:71-470).  The loops are host control flow only; the
model they drive enqueues HIP kernels.  Timing/log lines follow the reference protocol (warm-up
iterations excluded, "s / img", "s / class code")."""
import logging
import os
import time
from contextlib import ExitStack, contextmanager
from typing import Any, Dict, List

import torch
from torch import nn

from .distributed import get_world_size

logger = logging.getLogger(__name__)


@contextmanager
def inference_context(model: nn.Module):
    """detectron2.evaluation.inference_context: temporarily eval()."""
    mode = model.training
    model.eval()
    try:
        yield
    finally:
        model.train(mode)


def format_class_codes_shared(class_codes: List[Dict[str, Any]], device) -> Dict[str, torch.Tensor]:
    """The per-class records of loop A as one tensor per code field, row c = the class whose support_set_target is c
    (semantics of meta_learn_evaluation.py:71-103: fields other than "snnl", cls_bias flattened to (n,); an empty list is
    returned as it is; a class id nobody delivered fails in torch.cat like the reference's None slot does)."""
    n = len(class_codes)
    if n == 0:
        return class_codes
    fields = [f for f in class_codes[0]["class_code"] if f != "snnl"]
    by_target = {int(rec["support_set_target"]): rec["class_code"] for rec in class_codes}
    packed = {}
    for f in fields:
        rows = [by_target[c][f].to(device) if c in by_target and f in by_target[c] else None for c in range(n)]
        t = torch.cat(rows, dim=0)
        packed[f] = t.reshape(-1) if f == "cls_bias" else t
    if "snnl" in class_codes[0]["class_code"]:  # the reference keeps the key with its initial None slots and fails on it in cat
        raise TypeError("expected Tensor as element 0 in argument 0, but got NoneType")
    return packed


def inference_normalization(model, codes: List[Dict[str, Any]]):
    """meta_learn_evaluation.py:105-116: the run_type "meta_learn_normalize_code" call under eval mode and no_grad."""
    logger.info(f"Start normalizing class codes on {get_world_size()} devices")
    was_training = isinstance(model, nn.Module) and model.training
    if isinstance(model, nn.Module):
        model.eval()
    try:
        with torch.no_grad():
            return model(batched_inputs=None, class_code=codes, run_type="meta_learn_normalize_code")
    finally:
        if was_training:
            model.train(True)


def class_padded_hw(support_set, divisibility: int = 32):
    """(H, W) one class's support images are padded to when the class runs alone (ImageList.from_tensors: the maximum over its
    images, rounded up to the backbone's size divisibility)."""
    d = int(divisibility)
    if len(support_set) == 0:
        return (0, 0)
    h = max(int(rec["image"].shape[-2]) for rec in support_set)
    w = max(int(rec["image"].shape[-1]) for rec in support_set)
    return ((h + d - 1) // d * d, (w + d - 1) // d * d)


def _log_totals(kind: str, unit: str, total_time: float, compute_time: float, n: int, devices: int):
    logger.info("Total inference time: {:.3f}s ({:.6f} s / {} per device, on {} devices)".format(
        total_time, total_time / max(n, 1), unit, devices))
    logger.info("Total inference pure compute time: {:.3f}s ({:.6f} s / img per device, on {} devices)".format(
        compute_time, compute_time / max(n, 1), devices))


def inference_on_support_set_dataset(model, data_loader, output_dir: str = None) -> List[Dict[str, Any]]:
    """Loop A (meta_learn_evaluation.py:256-365): one item = one class {"support_set", "support_set_target",
    "class_name"}; returns [{support_set_target, class_name, class_code}] and optionally writes
    <output_dir>/<class_name>.pth (the file SylphPredictor reads, sylph/predictor.py:167-187)."""
    devices = get_world_size()
    total = len(data_loader)
    logger.info(f"Start generating class codes on {total} support sets on {devices} devices")
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
    num_warmup = min(5, max(total - 1, 0))
    start_time, compute = time.perf_counter(), 0.0
    results = []
    with ExitStack() as stack:
        if isinstance(model, nn.Module):
            stack.enter_context(inference_context(model))
        stack.enter_context(torch.no_grad())
        # Classes are grouped into batches of up to SYLPH_SUPPORT_BATCH support images (default 64) that share the backbone and
        # code-generator launches (model.forward_class_codes); the loader still yields -- and the model API still accepts --
        # one class per item, as in the reference.  Only classes with the same shot count AND the same padded size (see
        # class_padded_hw) share a batch, so every class sees exactly the tensors of its own one-class call.
        # SYLPH_SUPPORT_BATCH=0: one call per class.
        cap = int(os.environ.get("SYLPH_SUPPORT_BATCH", "64"))
        batched = cap > 0 and hasattr(model, "forward_class_codes")
        group: List[Any] = []

        def flush():
            nonlocal compute
            if not group:
                return
            t0 = time.perf_counter()
            if batched and len(group) > 1:
                codes = model.forward_class_codes([g for g in group])
            else:
                codes = [model(g, run_type="meta_learn_test_support") for g in group]
            codes = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in c.items()} for c in codes]  # device sync
            compute += time.perf_counter() - t0
            for g, c in zip(group, codes):
                result = {k: v for k, v in g[0].items() if k != "support_set"}
                result["class_code"] = c
                if output_dir is not None:
                    torch.save(result, os.path.join(output_dir, f"{result['class_name']}.pth"))
                results.append(result)
            group.clear()

        n_img = 0
        for idx, inputs in enumerate(data_loader):
            assert len(inputs) == 1, "inputs' batch size is not 1"
            if idx == num_warmup:
                flush()
                n_img = 0
                start_time, compute = time.perf_counter(), 0.0
            k = len(inputs[0]["support_set"])
            # A class may only join a group whose padded batch size equals the size the class would be padded to ALONE (the max
            # over ALL of its support images, rounded up to the size divisibility): the reference pads one class per call
            # (meta_one_stage_detector.py:229-254), and the activations next to the right / bottom border depend on that size.
            same = not group or (len(group[0][0]["support_set"]) == k and
                                 class_padded_hw(group[0][0]["support_set"]) == class_padded_hw(inputs[0]["support_set"]))
            if group and (not batched or not same or n_img + k > cap):
                flush()
                n_img = 0
            group.append(inputs)
            n_img += k
        flush()
    _log_totals("support", "class code", time.perf_counter() - start_time, compute, total - num_warmup, devices)
    return results


def inference_on_support_set_dataset_base(model, data_loader, all_id_map=None, base_id_map=None,
                                          output_dir: str = None) -> List[Dict[str, Any]]:
    """Base-class variant (meta_learn_evaluation.py:118-254): a class arrives in chunks of <= 10 shots
    carrying "len"/"total_len"; chunk codes are accumulated with weight len/total_len and the
    accumulated weight is kept in "acc_weight" for the cross-rank reduce."""
    from . import distributed as D
    rows, names, engine = [], {}, getattr(model, "engine", None)
    with ExitStack() as stack:
        if isinstance(model, nn.Module):
            stack.enter_context(inference_context(model))
        stack.enter_context(torch.no_grad())
        for inputs in data_loader:
            assert len(inputs) == 1, "inputs' batch size is not 1"
            code = model(inputs, run_type="meta_learn_test_support")  # device tensors; nothing is read back per chunk
            cid = int(inputs[0]["support_set_target"])
            names[cid] = inputs[0]["class_name"]
            weight = float(inputs[0]["len"]) / inputs[0]["total_len"]
            wn = code["cls_weight_norm"].reshape(1) * weight if "cls_weight_norm" in code else None
            rows.append(D.pack_codes(code["cls_conv"].reshape(1, 256) * weight, code["cls_bias"].reshape(1) * weight, [cid],
                                     [weight], wn, [names[cid]]))
    if not rows:
        return []
    packed = torch.cat(rows).contiguous()
    ncls = max(names) + 1
    # per-class accumulation of the weighted chunk codes in arrival order: ONE segmented reduce on the device
    # (sylph_reduce_codes, divide_by_acc = 0); acc_weight keeps the accumulated weight for the cross-rank reduce
    if engine is not None and packed.is_cuda:
        red = engine.reduce_codes(packed, ncls, divide_by_acc=False)
    else:
        red = D.scatter_by_class_id(D.reduce_packed_codes(packed, divide_by_acc=False), ncls)
    red = red.cpu()  # the only read-back of the loop
    results = []
    for cid in names:  # first-appearance order, as the reference's dict
        r = red[cid]
        cc = {"cls_conv": r[:256].reshape(1, 256, 1, 1).clone(), "cls_bias": r[256:257].reshape(1, 1, 1, 1).clone(),
              "acc_weight": float(r[D.F_ACC])}
        if float(r[D.F_HAS_WNORM]) > 0:
            cc["cls_weight_norm"] = r[D.F_WNORM:D.F_WNORM + 1].reshape(1, 1, 1, 1).clone()
        results.append({"support_set_target": cid, "class_name": names[cid], "class_code": cc})
    return results


class _NoOpEvaluator:
    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        return {}


def inference_on_dataset_with_class_codes(model, data_loader, evaluator, class_codes, cls_reweight=False,
                                          eval_with_pretrained_code=False):
    """Loop B (meta_learn_evaluation.py:367-470): model(inputs, class_code=..., run_type=
    "meta_learn_test_instance") per query batch, evaluator.process(inputs, outputs), evaluator.evaluate()."""
    devices = get_world_size()
    if eval_with_pretrained_code:
        assert class_codes is None  # meta_learn_evaluation.py:376-378: the model's own cls_logits are the codes
    else:
        assert class_codes is not None
    if cls_reweight:
        raise NotImplementedError("cls_reweight is not supported (CLS_REWEIGHT is False in every yaml)")
    total = len(data_loader)
    logger.info(f"Start inference with {'pretrained' if eval_with_pretrained_code else 'predicted'} class codes on {total} images")
    evaluator = evaluator if evaluator is not None else _NoOpEvaluator()
    evaluator.reset()
    num_warmup = min(5, max(total - 1, 0))
    start_time, compute = time.perf_counter(), 0.0
    with ExitStack() as stack:
        if isinstance(model, nn.Module):
            stack.enter_context(inference_context(model))
        stack.enter_context(torch.no_grad())
        for idx, inputs in enumerate(data_loader):
            if idx == num_warmup:
                start_time, compute = time.perf_counter(), 0.0
            t0 = time.perf_counter()
            outputs = model(inputs, class_code=class_codes, run_type="meta_learn_test_instance")
            # forward_instances ends on the count read-back: the device work of this batch is complete
            compute += time.perf_counter() - t0
            evaluator.process(inputs, outputs)
    _log_totals("query", "img", time.perf_counter() - start_time, compute, total - num_warmup, devices)
    results = evaluator.evaluate()
    return results if results is not None else {}


DET_ROW = 8  # image id | x0 | y0 | x1 | y1 | score | contiguous class id | valid


def detections_to_tensor(outputs: List[Dict[str, Any]], image_ids: List[int]) -> torch.Tensor:
    """Result sink, device side: every detection of a batch as one dense (n, 8) fp32 tensor on the device
    (image id, XYXY box, score, class, valid) -- no per-image / per-field .cpu() as in the d2 evaluators the reference
    calls from meta_learn_evaluation.py:428,465.  Image ids must be < 2^24 (exact in fp32)."""
    assert len(outputs) == len(image_ids)
    insts = [o["instances"] for o in outputs]
    dev = insts[0].pred_boxes.tensor.device if insts else torch.device("cpu")
    parts = []
    for img_id, i in zip(image_ids, insts):
        n = len(i)
        if n == 0:
            continue
        assert 0 <= int(img_id) < (1 << 24)
        parts.append(torch.cat([torch.full((n, 1), float(img_id), device=dev), i.pred_boxes.tensor.float(),
                                i.scores.float().unsqueeze(1), i.pred_classes.float().unsqueeze(1),
                                torch.ones(n, 1, device=dev)], dim=1))
    return torch.cat(parts) if parts else torch.zeros(0, DET_ROW, device=dev)


def gather_detection_rows(local: torch.Tensor, capacity: int) -> torch.Tensor:
    """Prediction gather across ranks (the reference pickles lists through comm.gather inside the evaluators): ONE
    all_gather_into_tensor of equal-size (capacity, 8) blocks, rank order preserved; rows with valid = 0 are padding.
    capacity >= the largest per-rank detection count (e.g. images per rank x POST_NMS_TOPK_TEST + ties)."""
    import torch.distributed as dist
    assert local.shape[0] <= capacity, f"{local.shape[0]} detections do not fit the gather block of {capacity}"
    block = torch.zeros(capacity, DET_ROW, dtype=torch.float32, device=local.device)
    block[: local.shape[0]] = local
    if not (dist.is_available() and dist.is_initialized()):
        return block
    out = torch.empty(dist.get_world_size() * capacity, DET_ROW, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, block)
    return out


def detection_rows_to_coco(rows: torch.Tensor, contiguous_to_dataset_id: Dict[int, int] = None) -> List[Dict[str, Any]]:
    """(n, 8) rows (ONE device->host copy here) -> COCO-json result dicts, boxes XYXY -> XYWH."""
    rows = rows.cpu()
    rows = rows[rows[:, 7] > 0].tolist()
    out = []
    for img_id, x0, y0, x1, y1, s, c, _ in rows:
        cid = int(c)
        out.append({"image_id": int(img_id), "category_id": contiguous_to_dataset_id[cid] if contiguous_to_dataset_id else cid,
                    "bbox": [x0, y0, x1 - x0, y1 - y0], "score": s})
    return out


def detections_to_coco_rows(outputs: List[Dict[str, Any]], image_ids: List[int],
                            contiguous_to_dataset_id: Dict[int, int] = None) -> List[Dict[str, Any]]:
    """COCO-json rows for a whole batch with ONE device->host copy (detections_to_tensor + detection_rows_to_coco)."""
    return detection_rows_to_coco(detections_to_tensor(outputs, image_ids), contiguous_to_dataset_id)
