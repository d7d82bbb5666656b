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
    has_acc = [float("acc_weight" in c["class_code"]) for c in codes]
    has_wn = all("cls_weight_norm" in c["class_code"] for c in codes)
    wn = torch.cat([c["class_code"]["cls_weight_norm"].reshape(1).float() for c in codes]).to(device) if has_wn else None
    return D.pack_codes(conv, bias, [int(c["support_set_target"]) for c in codes], acc, wn,
                        [c.get("class_name") for c in codes], has_acc=has_acc)


def _codes_from_rows(rows: torch.Tensor, keep_acc: Optional[bool], extras: Dict[int, Dict[str, Any]] = None) -> List[Dict[str, Any]]:
    """Valid packed rows (host) -> the reference's list-of-dicts form, in row order.  keep_acc None: a row gets its "acc_weight" key
    back iff the record it was packed from carried one (the explicit F_HAS_ACC lane, not a guess from the value)."""
    rows = rows.cpu()
    rows = rows[rows[:, D.F_VALID] > 0]
    names = D.unpack_names(rows)
    out = []
    for r, name in zip(rows, names):
        cid = int(round(float(r[D.F_CID])))
        cc = {"cls_conv": r[:256].reshape(1, 256, 1, 1).clone(), "cls_bias": r[256:257].reshape(1, 1, 1, 1).clone()}
        if float(r[D.F_HAS_WNORM]) > 0:
            cc["cls_weight_norm"] = r[D.F_WNORM:D.F_WNORM + 1].reshape(1, 1, 1, 1).clone()
        if keep_acc or (keep_acc is None and float(r[D.F_HAS_ACC]) > 0):
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
        carries (weights, bias, accumulated weight + whether the record had one, class id, weight norm, class name) travels in
        ONE dense fp32 block per rank through ONE all_gather_into_tensor over RCCL / gloo (sylph_amd.distributed): no pickle,
        no count exchange, no device read-back before the rows become host dicts again.  `capacity` = rows every rank reserves
        (`_episode` derives it from the loader's global length -- the InferenceSampler shard size ceil(n / world), known on every
        rank without communication); a bare call without it agrees on one with a scalar all_reduce(MAX) of the local counts.
        A rank holding more rows than `capacity` does not raise in front of the collective (the other ranks would hang in it): the
        overflow travels in the block and EVERY rank raises distributed.GatherOverflow afterwards."""
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
            capacity = max(int(capacity), 1)  # as gather_packed_codes / fit_block clamp it: a zero-row block cannot carry the overflow flag
            rows = D.gather_packed_codes(local, capacity).cpu()
            D.check_overflow(rows, capacity)  # same decision on every rank, after the collective
            # "acc_weight" comes back on exactly the records that carried it (explicit flag lane): nothing is inferred from values
            out_codes = _codes_from_rows(rows, keep_acc=None)
        else:
            out_codes = sub_class_codes
        if not reduce:
            return out_codes
        return reduce_class_code(out_codes, engine=engine)

    # ---- loaders: dataset registration is out of scope; a deployment overrides these three --------------------------
    def build_episodic_learning_detection_test_support_set_loader(self, cfg, dataset_name: str, seed: int = 0):
        """meta_fcos_runner.py:241-259 -> loop A's loader (one class per item) for (dataset, seed)."""
        raise NotImplementedError("dataset registration/loading (sylph/data/*) is out of scope: override this builder or pass "
                                  "support_loader / loaders= to _do_test_meta_learning")

    def build_episodic_learning_detection_test_support_set_base_loader(self, cfg, dataset_name: str):
        """meta_fcos_runner.py:261-279 -> the base-class all-ground-truth chunk loader (USE_ALL_GTS_IN_BASE_CLASSES)."""
        raise NotImplementedError("override build_episodic_learning_detection_test_support_set_base_loader")

    def build_episodic_learning_detection_test_query_loader(self, cfg, dataset_name: str):
        """meta_fcos_runner.py:281-296 -> loop B's loader."""
        raise NotImplementedError("override build_episodic_learning_detection_test_query_loader")

    def get_evaluator(self, cfg, dataset_name: str, output_folder: Optional[str] = None):
        """COCO / LVIS evaluators are out of scope; None = the no-op evaluator of inference_on_dataset_with_class_codes."""
        return None

    @staticmethod
    def _global_len(loader, fallback: Optional[int] = None) -> Optional[int]:
        """Items of a loader over ALL ranks (the synthetic loaders expose `num_items`; torch DataLoaders their dataset)."""
        for attr in ("num_items",):
            if hasattr(loader, attr):
                return int(getattr(loader, attr))
        ds = getattr(loader, "dataset", None)
        if ds is not None and hasattr(ds, "__len__"):
            return len(ds)
        return fallback

    def _episode(self, cfg, model, support_loader, query_loader, evaluator=None, base_support_loader=None,
                 output_folder: Optional[str] = None, num_classes: Optional[int] = None, eval_with_pretrained_code: bool = False):
        """ONE (dataset, seed) of meta_fcos_runner.py:497-560: support codes -> gather -> (base-class reduce + replace) ->
        normalise -> format -> query loop.  Returns (evaluator results, formatted class codes)."""
        class_codes = None
        if not eval_with_pretrained_code:
            sub = inference_on_support_set_dataset(model, support_loader, output_dir=output_folder)
            n_items = self._global_len(support_loader, num_classes)
            codes = self._gather_class_code(sub, capacity=D.shard_capacity(n_items) if n_items else None)
            if base_support_loader is not None:
                base_sub = inference_on_support_set_dataset_base(model, base_support_loader)
                # the base path delivers at most one accumulated row per class and rank
                n_base = num_classes if num_classes else self._global_len(base_support_loader)
                base = self._gather_class_code(base_sub, reduce=True, engine=getattr(model, "engine", None), capacity=n_base)
                by_cid = {int(c["support_set_target"]): c for c in base}
                codes = [dict(c, class_code=by_cid[int(c["support_set_target"])]["class_code"])
                         if int(c["support_set_target"]) in by_cid else c for c in codes]  # replace_class_code
            if str(cfg.MODEL.META_LEARN.CODE_GENERATOR.NAME) != "ROIEncoder":
                codes = inference_normalization(model, codes)  # ROIEncoder codes need none (and the reference call raises)
            if num_classes is not None:
                assert len(codes) == num_classes, \
                    f"Got {len(codes)} class codes for prediction, but expect to be {num_classes}."
            class_codes = format_class_codes_shared(codes, device=model.device)
        res = inference_on_dataset_with_class_codes(model, query_loader, evaluator, class_codes,
                                                    eval_with_pretrained_code=eval_with_pretrained_code)
        return res, class_codes

    def _do_test_meta_learning(self, cfg, model, support_loader=None, query_loader=None, evaluator=None, base_support_loader=None,
                               output_folder: Optional[str] = None, num_classes: Optional[int] = None, train_iter=None,
                               model_tag: str = "default", dataset_names: Optional[List[str]] = None):
        """meta_fcos_runner.py:451-672.

        * With explicit loaders (support_loader + query_loader): ONE dataset / seed; returns (results, class_codes) -- the form
          the episode tests and bench use.
        * Without: the reference's full loop -- for seed in range(TEST.REPEAT_TEST if final else 1), for dataset in
          DATASETS.TEST (or `dataset_names`): loaders from the build_* methods (seeded support sets, :497-503),
          EVAL_WITH_PRETRAINED_CODE on "base" datasets (:483-486), USE_ALL_GTS_IN_BASE_CLASSES (:507-518), then
          results[f"seed{s}"][dataset] per run, results[model_tag][dataset] = the mean over seeds of every "bbox" metric
          (:589-603) plus AP / APr / APc / APf "_avg" and "_std" over the seeds (:604-620); returns the results dict."""
        if support_loader is not None or query_loader is not None:
            return self._episode(cfg, model, support_loader, query_loader, evaluator, base_support_loader, output_folder, num_classes)
        names = list(dataset_names if dataset_names is not None else cfg.DATASETS.TEST)
        assert len(names)
        max_iter = cfg.get("SOLVER", {}).get("MAX_ITER", None) if hasattr(cfg, "get") else None
        is_final = train_iter is None or (max_iter is not None and train_iter == max_iter - 1)
        repeat = int(cfg.TEST.REPEAT_TEST) if is_final and "TEST" in cfg and "REPEAT_TEST" in cfg.TEST else 1
        results = OrderedDict()
        results[model_tag] = OrderedDict()
        main = D.get_rank() == 0
        for seed in range(repeat):
            logger.info(f"{seed} out of {repeat} tests.")
            results[f"seed{seed}"] = OrderedDict()
            for name in names:
                pretrained = "base" in name and bool(cfg.MODEL.META_LEARN.EVAL_WITH_PRETRAINED_CODE)
                folder = None
                if output_folder or cfg.get("OUTPUT_DIR", None):
                    folder = os.path.join(output_folder or cfg.OUTPUT_DIR, "inference", model_tag,
                                          str(train_iter) if train_iter is not None else "final", name, str(seed))
                sup = base = None
                if not pretrained:
                    sup = self.build_episodic_learning_detection_test_support_set_loader(cfg, name, seed)
                    if bool(cfg.MODEL.META_LEARN.USE_ALL_GTS_IN_BASE_CLASSES):
                        base = self.build_episodic_learning_detection_test_support_set_base_loader(cfg, name)
                qry = self.build_episodic_learning_detection_test_query_loader(cfg, name)
                ev = self.get_evaluator(cfg, name, output_folder=folder)
                n_cls = self._global_len(sup) if sup is not None else None
                per, _ = self._episode(cfg, model, sup, qry, ev, base, folder, n_cls, eval_with_pretrained_code=pretrained)
                if not main:
                    continue
                results[f"seed{seed}"][name] = per
                bbox = per.get("bbox") if isinstance(per, dict) else None
                if seed == 0:
                    results[model_tag][name] = {k: (dict(v) if isinstance(v, dict) else v) for k, v in per.items()} \
                        if isinstance(per, dict) else per
                elif bbox is not None:
                    acc = results[model_tag][name]["bbox"]
                    for k in acc:
                        acc[k] += bbox[k]
                        if seed == repeat - 1:
                            acc[k] /= repeat
                if is_final and seed == repeat - 1 and bbox is not None:
                    import numpy as np
                    for k in ("AP", "APr", "APc", "APf"):
                        hist = [results[f"seed{s}"][name]["bbox"][k] for s in range(repeat) if k in results[f"seed{s}"][name]["bbox"]]
                        if hist:
                            results[model_tag][name]["bbox"][f"{k}_avg"] = float(np.array(hist).mean())
                            results[model_tag][name]["bbox"][f"{k}_std"] = float(np.array(hist).std())
        return results

    def do_test(self, cfg, model, train_iter=None, support_loader=None, query_loader=None, evaluator=None):
        """meta_fcos_runner.py:674-701.  With explicit episodic loaders: one episode (sylph_amd.data has synthetic ones emitting
        the reference's item shapes).  Without: the multi-seed / multi-dataset loop over the build_* loader methods."""
        if not cfg.MODEL.META_LEARN.EPISODIC_LEARNING:
            raise NotImplementedError("base-detector evaluation is out of scope")
        if support_loader is None and query_loader is None:
            return self._do_test_meta_learning(cfg, model, train_iter=train_iter)
        if support_loader is None or query_loader is None:
            raise NotImplementedError(
                "dataset registration/loading (sylph/data/*) is out of scope: pass support_loader and query_loader")
        res, _ = self._do_test_meta_learning(cfg, model, support_loader, query_loader, evaluator)
        return OrderedDict(default=res)


class MetaFCOSROIEncoderRunner(MetaFCOSRunner):
    """sylph/runner/meta_fcos_roi_encoder_runner.py: ROIEncoder config defaults."""

    def get_default_cfg(self):
        return _config.get_roi_encoder_default_cfg()
