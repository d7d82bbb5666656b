"""Test-only import shim: lets the reference's Sylph-owned hot-path modules import in the build
container, where detectron2 / adet / fvcore / d2go / pycocotools are absent.

Used ONLY by tests/golden/gen_goldens.py to generate golden vectors from the reference itself.
Nothing here ships to the GPU box as part of the product and nothing in the product imports it.
The stand-ins are written from scratch; the "live" third-party primitives (ROIPooler, ml_nms,
compute_locations) are plugged with this repo's own oracle restatements, so a golden that passes
through them pins the Sylph-owned arithmetic around them, not the primitives themselves.
"""
import contextlib
import sys
import types
from typing import Any, Dict, List, Tuple

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------- containers
class Boxes:
    def __init__(self, tensor):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        self.tensor = tensor

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def to(self, device):
        return Boxes(self.tensor.to(device))

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def cat(lst):
        return Boxes(torch.cat([b.tensor for b in lst], dim=0))


class Instances:
    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(name)
        return self._fields[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def remove(self, name):
        del self._fields[name]

    def to(self, *a, **k):
        ret = Instances(self._image_size)
        for n, v in self._fields.items():
            ret.set(n, v.to(*a, **k) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for n, v in self._fields.items():
            ret.set(n, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    @staticmethod
    def cat(lst):
        ret = Instances(lst[0].image_size)
        for k in lst[0]._fields.keys():
            vals = [i.get(k) for i in lst]
            if isinstance(vals[0], torch.Tensor):
                vals = torch.cat(vals, dim=0)
            elif isinstance(vals[0], Boxes):
                vals = Boxes.cat(vals)
            ret.set(k, vals)
        return ret


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class Registry:
    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


# ----------------------------------------------------------------------------- live primitives
class ROIPooler(nn.Module):
    """Stand-in with the detectron2 constructor; arithmetic = this repo's oracle restatement."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        import math
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else output_size
        self.scales = scales
        self.min_level = int(round(-math.log2(scales[0])))
        self.max_level = int(round(-math.log2(scales[-1])))
        self.canonical_box_size, self.canonical_level = canonical_box_size, canonical_level
        assert pooler_type == "ROIAlignV2" and sampling_ratio == 0
        self.level_poolers = nn.ModuleList([_LevelROIAlign(self.output_size[0], sc) for sc in scales])

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]):
        from oracle.roi_align import roi_pooler
        assert all(len(b) == 1 for b in box_lists), "shim supports one box per image"
        boxes = torch.cat([b.tensor for b in box_lists], dim=0)
        strides = [int(round(1.0 / s)) for s in self.scales]
        return roi_pooler([f.detach() for f in x], boxes, strides, self.output_size[0])


class _LevelROIAlign(nn.Module):
    """detectron2.layers.ROIAlign(aligned=True, sampling_ratio=0) on (K,5) rois, via the oracle restatement."""

    def __init__(self, out_size, scale):
        super().__init__()
        self.out_size, self.scale = out_size, scale

    def forward(self, x, rois):
        from oracle.roi_align import roi_align_single
        outs = [roi_align_single(x[int(r[0])].detach(), r[1:].tolist(), self.scale, self.out_size) for r in rois]
        if not outs:
            return x.new_zeros((0, x.shape[1], self.out_size, self.out_size))
        return torch.stack(outs, dim=0)


def convert_boxes_to_pooler_format(box_lists):
    rows = []
    for i, b in enumerate(box_lists):
        t = b.tensor
        rows.append(torch.cat([torch.full((len(t), 1), float(i)), t], dim=1))
    return torch.cat(rows, dim=0)


def assign_boxes_to_levels_shim(box_lists, min_level, max_level, canonical_box_size, canonical_level):
    from oracle.roi_align import assign_boxes_to_levels
    boxes = torch.cat([b.tensor for b in box_lists], dim=0)
    return assign_boxes_to_levels(boxes, min_level, max_level, canonical_box_size, canonical_level)


def ml_nms(boxlist, nms_thresh, max_proposals=-1, score_field="scores", label_field="labels"):
    from oracle.decode import nms_per_class
    if nms_thresh <= 0:
        return boxlist
    keep = nms_per_class(boxlist.pred_boxes.tensor.numpy(), boxlist.scores.numpy(),
                         boxlist.pred_classes.numpy(), nms_thresh)
    keep = torch.from_numpy(keep)
    if max_proposals > 0:
        keep = keep[:max_proposals]
    return boxlist[keep]


def compute_locations(h, w, stride, device):
    from oracle.decode import compute_locations as cl
    return cl(h, w, stride).to(device)


# ----------------------------------------------------------------------------- inert modules
class _Inert(types.ModuleType):
    """Module whose every attribute is an inert placeholder class/callable."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        ph = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, ph)
        return ph


class _PathManager:
    @staticmethod
    def exists(p):
        return False

    @staticmethod
    def get_local_path(p):
        return p


def _configurable(init_func=None, *, from_config=None):
    """detectron2.config.configurable: Class(cfg, ...) -> Class(**Class.from_config(cfg, ...))."""
    import functools

    def wrap(orig_init):
        @functools.wraps(orig_init)
        def wrapped(self, *args, **kwargs):
            fc = type(self).from_config
            first = args[0] if args else kwargs.get("cfg")
            if first is not None and hasattr(first, "MODEL"):
                orig_init(self, **fc(*args, **kwargs))
            else:
                orig_init(self, *args, **kwargs)
        return wrapped
    if init_func is not None:
        return wrap(init_func)
    return wrap


class Conv2dWrapper(nn.Conv2d):
    """detectron2.layers.Conv2d: nn.Conv2d with optional ``norm`` and ``activation`` submodules."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = nn.functional.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    assert norm == "GN", norm
    return nn.GroupNorm(32, out_channels)


def install():
    """Register stand-ins in sys.modules and put the reference on sys.path."""
    def mod(name, inert=False, **attrs):
        m = _Inert(name) if inert else types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], child, m)
        return m

    def cat(tensors, dim=0):
        return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)

    def nonzero_tuple(x):
        return x.nonzero(as_tuple=True)

    mod("detectron2", inert=True)
    mod("detectron2.utils", inert=True)
    mod("detectron2.utils.registry", Registry=Registry)
    mod("detectron2.utils.file_io", PathManager=_PathManager)
    mod("detectron2.utils.comm", inert=True, get_world_size=lambda: 1)
    mod("detectron2.utils.logger", inert=True)
    mod("detectron2.layers", inert=True, ShapeSpec=ShapeSpec, cat=cat, nonzero_tuple=nonzero_tuple,
        Conv2d=Conv2dWrapper, get_norm=get_norm)
    mod("detectron2.layers.batch_norm", inert=True)
    mod("detectron2.structures", inert=True, Boxes=Boxes, Instances=Instances)
    mod("detectron2.modeling", inert=True)
    mod("detectron2.modeling.poolers", inert=True, ROIPooler=ROIPooler,
        assign_boxes_to_levels=assign_boxes_to_levels_shim,
        convert_boxes_to_pooler_format=convert_boxes_to_pooler_format)
    mod("detectron2.modeling.proposal_generator", inert=True)
    mod("detectron2.modeling.proposal_generator.build", PROPOSAL_GENERATOR_REGISTRY=Registry("PG"))
    mod("detectron2.config", inert=True, configurable=_configurable)
    mod("detectron2.data", inert=True)
    mod("detectron2.evaluation", inert=True)
    mod("detectron2.evaluation.coco_evaluation", inert=True)
    mod("detectron2.evaluation.evaluator", inert=True, inference_context=contextlib.nullcontext)
    mod("detectron2.evaluation.fast_eval_api", inert=True)
    mod("adet", inert=True)
    mod("adet.layers", inert=True, ml_nms=ml_nms)
    mod("adet.utils", inert=True)
    mod("adet.utils.comm", inert=True, compute_locations=compute_locations)
    mod("fvcore", inert=True)
    mod("fvcore.nn", inert=True)
    mod("fvcore.nn.weight_init", c2_msra_fill=lambda m: None, c2_xavier_fill=lambda m: None)
    sys.modules["fvcore.nn"].weight_init = sys.modules["fvcore.nn.weight_init"]
    mod("pycocotools", inert=True)
    mod("pycocotools.cocoeval", inert=True)
    mod("pycocotools.coco", inert=True)
    mod("d2go", inert=True)
    mod("d2go.utils", inert=True)
    mod("d2go.utils.misc", inert=True)
    # sylph.data.* needs datasets/lvis on disk: only the names are needed at import time
    mod("sylph.data", inert=True)
    mod("sylph.data.data_injection", inert=True)
    mod("sylph.data.data_injection.classes", inert=True)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class Cfg(dict):
    """Attribute dict standing in for a yacs CfgNode."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(d):
    if isinstance(d, dict):
        return Cfg({k: to_cfg(v) for k, v in d.items()})
    return d
