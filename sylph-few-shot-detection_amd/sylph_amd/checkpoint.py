"""Checkpoint readers for the weights the reference's configs point at.

`DetectionCheckpointer(model).load(cfg.MODEL.WEIGHTS)` (sylph/predictor.py:87-88, sylph/runner/meta_fcos_runner.py:232-288)
accepts three families of files; all end up as the reference state-dict keys of SURVEY.md 8b:

  * a Sylph / detectron2 training checkpoint `model_final.pth`: torch.save({"model": state_dict, "iteration": ..., ...});
  * a detectron2 model-zoo pickle (`*.pkl`: {"model": {name: ndarray}, "__author__": ..., "matching_heuristics": True}) with
    torch-style names (`stem.conv1.weight`, `res2.0.conv1.norm.weight`, ...) that lack the `backbone.bottom_up.` prefix;
  * the MSRA ImageNet backbones the yamls name (`detectron2://ImageNetPretrained/MSRA/R-50.pkl`, Base-FCOS.yaml): Caffe2 blob
    names (`conv1_w`, `res_conv1_bn_s`, `res2_0_branch2a_w`, `res2_0_branch2a_bn_b`, `res2_0_branch1_w`, `fc1000_w`) with
    the BatchNorm statistics already absorbed into an affine (`_bn_s`, `_bn_b`): detectron2's c2_model_loading renames them
    and FrozenBatchNorm2d fills running_mean = 0, running_var = 1 for the missing statistics.
"""
import pickle
import re
from typing import Any, Dict

import numpy as np
import torch

_BRANCH = {"branch2a": "conv1", "branch2b": "conv2", "branch2c": "conv3", "branch1": "shortcut"}


def _to_tensor(v: Any) -> Any:
    if isinstance(v, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(v))
    return v


def read_file(path: str) -> Dict[str, Any]:
    """The raw name -> array mapping stored in a .pth / .pkl checkpoint."""
    if str(path).endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
    else:
        data = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(data, dict):
        for key in ("model", "blobs", "state_dict"):
            if key in data and isinstance(data[key], dict):
                data = data[key]
                break
    if not isinstance(data, dict):
        raise ValueError(f"{path}: not a checkpoint dictionary")
    return {k: _to_tensor(v) for k, v in data.items() if not k.startswith("__")}


def is_caffe2_names(sd: Dict[str, Any]) -> bool:
    return any(k in sd for k in ("conv1_w", "res_conv1_bn_s")) or any(re.match(r"res\d+_\d+_branch", k) for k in sd)


def convert_caffe2_names(sd: Dict[str, Any]) -> Dict[str, Any]:
    """detectron2 c2_model_loading.convert_basic_c2_names for the ResNet blobs (backbone only; fc1000 / momentum blobs dropped)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("_momentum") or k.startswith("fc1000") or k.startswith("pred_") or k in ("lr", "weight_order"):
            continue
        m = re.match(r"^res(\d+)_(\d+)_(branch2a|branch2b|branch2c|branch1)_(w|b|bn_s|bn_b|bn_rm|bn_riv)$", k)
        if m:
            base = f"res{m.group(1)}.{int(m.group(2))}.{_BRANCH[m.group(3)]}"
        elif re.match(r"^conv1_(w|b)$", k) or re.match(r"^res_conv1_(bn_s|bn_b|bn_rm|bn_riv)$", k):
            base = "stem.conv1"
            m = re.match(r"^(?:res_)?conv1_(w|b|bn_s|bn_b|bn_rm|bn_riv)$", k)
        else:
            continue  # not a backbone blob
        kind = m.group(m.lastindex)
        # bn_riv goes to running_var UNCHANGED: detectron2's convert_basic_c2_names only renames it (c2_model_loading.py), and the
        # reference loads checkpoints through that loader
        suffix = {"w": "weight", "b": "bias", "bn_s": "norm.weight", "bn_b": "norm.bias", "bn_rm": "norm.running_mean",
                  "bn_riv": "norm.running_var"}[kind]
        out[f"{base}.{suffix}"] = v
    return out


def to_reference_keys(sd: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """Any of the three families -> reference state-dict keys, FrozenBN statistics completed."""
    if is_caffe2_names(sd):
        sd = convert_caffe2_names(sd)
    out = {}
    for k, v in sd.items():
        if not torch.is_tensor(v):
            continue
        if re.match(r"^(stem|res\d+)\.", k):  # backbone-only weights (model zoo / converted MSRA): detectron2 matches by suffix
            k = "backbone.bottom_up." + k
        elif k.startswith("bottom_up."):
            k = "backbone." + k
        out[k] = v
    for k in [k for k in out if k.endswith(".norm.weight") and ".bottom_up." in k]:  # FrozenBatchNorm2d._load_from_state_dict
        p = k[: -len("weight")]
        out.setdefault(p + "running_mean", torch.zeros_like(out[k]))
        out.setdefault(p + "running_var", torch.ones_like(out[k]))
    return out


def load_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
    return to_reference_keys(read_file(path))
