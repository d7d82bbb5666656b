"""MetaFCOSRunner: the reference's orchestration/plugin point for this path
(sylph/runner/meta_fcos_runner.py:92-701), reduced to inference: get_default_cfg, build_model,
_gather_class_code and the meta-test control flow of _do_test_meta_learning.  Dataset registration,
evaluators and training are out of scope: loaders are passed in (or synthetic)."""
import importlib
import logging
import os
from collections import OrderedDict
from typing import Any, Dict, List, Optional

import torch

from . import config as _config
from . import distributed as D
from .evaluation import (format_class_codes_shared, inference_normalization, inference_on_dataset_with_class_codes,
                         inference_on_support_set_dataset, inference_on_support_set_dataset_base)
from .modeling import build_model as _build_model

logger = logging.getLogger(__name__)


def create_runner(class_full_name: str, *args, **kwargs):
    """d2go.runner.create_runner: dotted path -> instance ("sylph_amd.runner.MetaFCOSRunner"; the
    reference's own "sylph.runner.MetaFCOSRunner" is accepted and rerouted here)."""
    if class_full_name.startswith("sylph.runner."):
        class_full_name = "sylph_amd.runner." + class_full_name[len("sylph.runner."):]
    module_name, _, cls_name = class_full_name.rpartition(".")
    cls = getattr(importlib.import_module(module_name), cls_name)
    return cls(*args, **kwargs)


def create_cfg(default_cfg, config_file: Optional[str], overwrite_opts: Optional[List[Any]] = None):
    """tools/setup.py:190-209 create_cfg_from_cli_args core: defaults <- yaml (sylph:// ok) <- opts."""
    cfg = default_cfg.clone()
    if config_file:
        cfg.merge_from_file(config_file)
    if overwrite_opts:
        cfg.merge_from_list(overwrite_opts)
    return cfg


def _rows_from_codes(codes: List[Dict[str, Any]], device) -> torch.Tensor:
    """list of {"support_set_target", "class_name", "class_code": {...}} -> packed rows (sylph_amd.distributed layout)."""
    if not codes:
        return torch.zeros(0, D.ROW, device=device)
    conv = torch.cat([c["class_code"]["cls_conv"].reshape(1, 256).float() for c in codes]).to(device)
    bias = torch.cat([c["class_code"]["cls_bias"].reshape(1).float() for c in codes]).to(device)
    acc = [float(c["class_code"].get("acc_weight", 1.0)) for c in codes]
    has_wn = all("cls_weight_norm" in c["class_code"] for c in codes)
    wn = torch.cat([c["class_code"]["cls_weight_norm"].reshape(1).float() for c in codes]).to(device) if has_wn else None
    return D.pack_codes(conv, bias, [int(c["support_set_target"]) for c in codes], acc, wn,
                        [c.get("class_name") for c in codes])


def _codes_from_rows(rows: torch.Tensor, keep_acc: bool, extras: Dict[int, Dict[str, Any]] = None) -> List[Dict[str, Any]]:
    """Valid packed rows (host) -> the reference's list-of-dicts form, in row order."""
    rows = rows.cpu()
    rows = rows[rows[:, D.F_VALID] > 0]
    names = D.unpack_names(rows)
    out = []
    for r, name in zip(rows, names):
        cid = int(round(float(r[D.F_CID])))
        cc = {"cls_conv": r[:256].reshape(1, 256, 1, 1).clone(), "cls_bias": r[256:257].reshape(1, 1, 1, 1).clone()}
        if float(r[D.F_HAS_WNORM]) > 0:
            cc["cls_weight_norm"] = r[D.F_WNORM:D.F_WNORM + 1].reshape(1, 1, 1, 1).clone()
        if keep_acc:
            cc["acc_weight"] = float(r[D.F_ACC])
        rec = dict(extras.get(cid, {})) if extras else {}
        rec.update({"support_set_target": cid, "class_name": rec.get("class_name") or name, "class_code": cc})
        out.append(rec)
    return out


def reduce_class_code(out_codes: List[Dict], engine=None) -> List[Dict]:
    """sylph/modeling/code_generator/utils.py:397-427 on the dict form.  With an Engine the sums run on the GPU
    (sylph_reduce_codes, fixed row order); without one (CPU-only unit tests) on host rows with the same arithmetic."""
    if len(out_codes) == 0:
        return out_codes
    assert "class_code" in out_codes[0]
    others = {}
    for c in out_codes:
        others.setdefault(int(c["support_set_target"]), {k: v for k, v in c.items() if k != "class_code"})
    if engine is not None:
        rows = _rows_from_codes(out_codes, engine.device)
        ncls = max(others) + 1
        red = engine.reduce_codes(rows.contiguous(), ncls).cpu()
        first = list(dict.fromkeys(int(c["support_set_target"]) for c in out_codes))  # the reference keeps first-appearance order
        red = red[torch.tensor(first, dtype=torch.long)]
    else:
        red = D.reduce_packed_codes(_rows_from_codes(out_codes, torch.device("cpu")))
    return _codes_from_rows(red, keep_acc=False, extras=others)


class MetaFCOSRunner:
    def __init__(self):
        self._logger = logging.getLogger(__name__)

    def get_default_cfg(self):
        """meta_fcos_runner.py:104-114."""
        return _config.get_default_cfg()

    def build_model(self, cfg, eval_only: bool = False, dtype: Optional[str] = None):
        """d2go GeneralizedRCNNRunner.build_model: registry lookup + optional MODEL.WEIGHTS load."""
        model = _build_model(cfg, dtype=dtype)
        if cfg.MODEL.WEIGHTS and os.path.exists(str(cfg.MODEL.WEIGHTS)):
            model.load_checkpoint(str(cfg.MODEL.WEIGHTS))
        if eval_only:
            model.eval()
        return model

    @classmethod
    def _gather_class_code(cls, sub_class_codes: List[Dict[str, Any]], reduce: bool = False, capacity: Optional[int] = None,
                           engine=None) -> List[Dict[str, Any]]:
        """meta_fcos_runner.py:381-439.  Same result as all_gather_object + rank-order flatten, but everything a code
        carries (weights, bias, accumulated weight, class id, weight norm, class name) travels in ONE dense fp32 block
        per rank through ONE all_gather_into_tensor over RCCL / gloo (sylph_amd.distributed): no pickle, no count
        exchange.  `capacity` = rows every rank reserves (default: the InferenceSampler shard size is not known here, so
        the maximum over ranks is agreed on by one scalar all_reduce; callers that know it pass it and skip that)."""
        world = D.get_world_size()
        if world > 1:
            import torch.distributed as dist
            dev = sub_class_codes[0]["class_code"]["cls_conv"].device if sub_class_codes else torch.device("cpu")
            if dist.get_backend() == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
            local = _rows_from_codes(sub_class_codes, dev)
            if capacity is None:
                cap = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
                dist.all_reduce(cap, op=dist.ReduceOp.MAX)
                capacity = max(int(cap.item()), 1)
            rows = D.gather_packed_codes(local, capacity)
            # decided on the GATHERED rows, i.e. identically on every rank (a rank with an empty shard has no local evidence)
            valid = rows[:, D.F_VALID] > 0
            has_acc = reduce or bool(((rows[:, D.F_ACC] != 1.0) & valid).any().item())
            out_codes = _codes_from_rows(rows, keep_acc=has_acc)
        else:
            out_codes = sub_class_codes
        if not reduce:
            return out_codes
        return reduce_class_code(out_codes, engine=engine)

    def _do_test_meta_learning(self, cfg, model, support_loader, query_loader, evaluator=None, base_support_loader=None,
                               output_folder: Optional[str] = None, num_classes: Optional[int] = None):
        """Control flow of meta_fcos_runner.py:451-560 for ONE dataset/seed: support codes -> gather ->
        (base-class reduce + replace) -> normalise -> format -> query loop."""
        sub = inference_on_support_set_dataset(model, support_loader, output_dir=output_folder)
        codes = self._gather_class_code(sub, capacity=D.shard_capacity(num_classes) if num_classes else None)
        if base_support_loader is not None:
            base_sub = inference_on_support_set_dataset_base(model, base_support_loader)
            base = self._gather_class_code(base_sub, reduce=True, engine=getattr(model, "engine", None))
            by_cid = {int(c["support_set_target"]): c for c in base}
            codes = [dict(c, class_code=by_cid[int(c["support_set_target"])]["class_code"])
                     if int(c["support_set_target"]) in by_cid else c for c in codes]
        if str(cfg.MODEL.META_LEARN.CODE_GENERATOR.NAME) != "ROIEncoder":
            codes = inference_normalization(model, codes)  # ROIEncoder codes need none (and the reference call raises)
        if num_classes is not None:
            assert len(codes) == num_classes, \
                f"Got {len(codes)} class codes for prediction, but expect to be {num_classes}."
        class_codes = format_class_codes_shared(codes, device=model.device)
        return inference_on_dataset_with_class_codes(model, query_loader, evaluator, class_codes), class_codes

    def do_test(self, cfg, model, train_iter=None, support_loader=None, query_loader=None, evaluator=None):
        """meta_fcos_runner.py:674-701.  Dataset-backed loaders are out of scope; pass episodic loaders
        (sylph_amd.data has synthetic ones emitting the reference's item shapes)."""
        if not cfg.MODEL.META_LEARN.EPISODIC_LEARNING:
            raise NotImplementedError("base-detector evaluation is out of scope")
        if support_loader is None or query_loader is None:
            raise NotImplementedError(
                "dataset registration/loading (sylph/data/*) is out of scope: pass support_loader and query_loader")
        res, _ = self._do_test_meta_learning(cfg, model, support_loader, query_loader, evaluator)
        return OrderedDict(default=res)


class MetaFCOSROIEncoderRunner(MetaFCOSRunner):
    """sylph/runner/meta_fcos_roi_encoder_runner.py: ROIEncoder config defaults."""

    def get_default_cfg(self):
        return _config.get_roi_encoder_default_cfg()
