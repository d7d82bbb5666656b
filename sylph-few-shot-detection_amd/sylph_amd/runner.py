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


def reduce_class_code(out_codes: List[Dict]) -> List[Dict]:
    """sylph/modeling/code_generator/utils.py:397-427 on the dict form (via the packed-row reducer)."""
    if len(out_codes) == 0:
        return out_codes
    assert "class_code" in out_codes[0]
    others = {}
    for c in out_codes:
        others.setdefault(int(c["support_set_target"]), {k: v for k, v in c.items() if k != "class_code"})
    conv = torch.cat([c["class_code"]["cls_conv"].reshape(1, 256).float() for c in out_codes])
    bias = torch.cat([c["class_code"]["cls_bias"].reshape(1).float() for c in out_codes])
    rows = D.pack_codes(conv, bias, [int(c["support_set_target"]) for c in out_codes],
                        [float(c["class_code"]["acc_weight"]) for c in out_codes])
    red = D.reduce_packed_codes(rows)
    results = []
    for r in red:
        cid = int(round(float(r[D.F_CID])))
        rec = dict(others[cid])
        rec["class_code"] = {"cls_conv": r[:256].reshape(1, 256, 1, 1).clone(), "cls_bias": r[256:257].reshape(1, 1, 1, 1).clone()}
        results.append(rec)
    return results


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
    def _gather_class_code(cls, sub_class_codes: List[Dict[str, Any]], reduce: bool = False) -> List[Dict[str, Any]]:
        """meta_fcos_runner.py:381-439.  Same result as all_gather_object + rank-order flatten, but the
        codes travel as ONE dense fp32 block per rank (sylph_amd.distributed) over RCCL/gloo; class names
        (host metadata) ride along through a small object gather only when world_size > 1."""
        world = D.get_world_size()
        if world > 1:
            import torch.distributed as dist
            dev = sub_class_codes[0]["class_code"]["cls_conv"].device if sub_class_codes else torch.device("cpu")
            if dist.get_backend() == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
            if sub_class_codes:
                conv = torch.cat([c["class_code"]["cls_conv"].reshape(1, 256).float() for c in sub_class_codes]).to(dev)
                bias = torch.cat([c["class_code"]["cls_bias"].reshape(1).float() for c in sub_class_codes]).to(dev)
                acc = [float(c["class_code"].get("acc_weight", 1.0)) for c in sub_class_codes]
                local = D.pack_codes(conv, bias, [int(c["support_set_target"]) for c in sub_class_codes], acc)
            else:
                local = torch.zeros(0, D.ROW, device=dev)
            rows = D.gather_packed_codes(local).cpu()
            names = [None] * world
            dist.all_gather_object(names, [(int(c["support_set_target"]), c.get("class_name")) for c in sub_class_codes])
            flat_names = [n for sub in names for n in sub]
            has_acc = any("acc_weight" in c["class_code"] for c in sub_class_codes) or reduce
            out_codes = []
            for r, (cid, name) in zip(rows, flat_names):
                cc = {"cls_conv": r[:256].reshape(1, 256, 1, 1).clone(), "cls_bias": r[256:257].reshape(1, 1, 1, 1).clone()}
                if has_acc:
                    cc["acc_weight"] = float(r[D.F_ACC])
                out_codes.append({"support_set_target": cid, "class_name": name, "class_code": cc})
        else:
            out_codes = sub_class_codes
        if not reduce:
            return out_codes
        return reduce_class_code(out_codes)

    def _do_test_meta_learning(self, cfg, model, support_loader, query_loader, evaluator=None, base_support_loader=None,
                               output_folder: Optional[str] = None, num_classes: Optional[int] = None):
        """Control flow of meta_fcos_runner.py:451-560 for ONE dataset/seed: support codes -> gather ->
        (base-class reduce + replace) -> normalise -> format -> query loop."""
        sub = inference_on_support_set_dataset(model, support_loader, output_dir=output_folder)
        codes = self._gather_class_code(sub)
        if base_support_loader is not None:
            base_sub = inference_on_support_set_dataset_base(model, base_support_loader)
            base = self._gather_class_code(base_sub, reduce=True)
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
