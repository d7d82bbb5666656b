"""Minimal containers with the detectron2 surface the Sylph inference path touches
(detectron2.structures.{Boxes, Instances}; call sites: sylph/modeling/meta_fcos/fcos_outputs.py:999-1006,
sylph/modeling/code_generator/utils.py:27-47, sylph/predictor.py:269-274).  Written from the public
behaviour, not copied: only what callers of the path use is provided."""
from typing import Any, Dict, Iterator, List, Tuple, Union

import torch


class Boxes:
    """(N, 4) XYXY absolute boxes."""

    def __init__(self, tensor: torch.Tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self) -> "Boxes":
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs) -> "Boxes":
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self) -> torch.Tensor:
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold: float = 0.0) -> torch.Tensor:
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def __getitem__(self, item) -> "Boxes":
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def __len__(self) -> int:
        return self.tensor.shape[0]

    def __iter__(self) -> Iterator[torch.Tensor]:
        yield from self.tensor

    def __repr__(self) -> str:
        return "Boxes(" + str(self.tensor) + ")"

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def cat(boxes_list: List["Boxes"]) -> "Boxes":
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0))


class Instances:
    """Per-image container of equally long fields (pred_boxes, scores, pred_classes, locations,
    fpn_levels for detections; gt_boxes, gt_classes for support annotations)."""

    def __init__(self, image_size: Tuple[int, int], **kwargs: Any):
        self._image_size = image_size
        self._fields: Dict[str, Any] = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self) -> Tuple[int, int]:
        return self._image_size

    def __setattr__(self, name: str, val: Any) -> None:
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name: str) -> Any:
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name: str, value: Any) -> None:
        if len(self._fields):
            assert len(self) == len(value), f"Adding a field of length {len(value)} to Instances of length {len(self)}"
        self._fields[name] = value

    def has(self, name: str) -> bool:
        return name in self._fields

    def remove(self, name: str) -> None:
        del self._fields[name]

    def get(self, name: str) -> Any:
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *args, **kwargs) -> "Instances":
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(*args, **kwargs) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item: Union[int, slice, torch.Tensor]) -> "Instances":
        if isinstance(item, int):
            item = slice(item, None, len(self)) if item >= 0 else slice(item, item + 1 or None)
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self) -> int:
        for v in self._fields.values():
            return len(v)
        return 0

    @staticmethod
    def cat(instance_lists: List["Instances"]) -> "Instances":
        assert len(instance_lists) > 0
        ret = Instances(instance_lists[0].image_size)
        for k in instance_lists[0]._fields.keys():
            vals = [i.get(k) for i in instance_lists]
            if isinstance(vals[0], torch.Tensor):
                vals = torch.cat(vals, dim=0)
            elif hasattr(type(vals[0]), "cat"):
                vals = type(vals[0]).cat(vals)
            ret.set(k, vals)
        return ret

    def __repr__(self) -> str:
        s = f"Instances(num_instances={len(self)}, image_height={self._image_size[0]}, image_width={self._image_size[1]}, "
        return s + "fields=[" + ", ".join(f"{k}: {v}" for k, v in self._fields.items()) + "])"
