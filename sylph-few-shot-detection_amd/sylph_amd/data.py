"""Synthetic episodic loaders that emit exactly the item shapes the reference's loaders do
(sylph/data/build.py:239-282,578-592,749-763; mapper sylph/data/dataset_mapper/
meta_learn_dataset_mapper.py:230-256), sharded across ranks in contiguous blocks like
detectron2's InferenceSampler.  Real dataset registration / image decoding is out of scope."""
from typing import Dict, List

import torch

from .distributed import inference_shard
from .structures import Boxes, Instances


def _image(h, w, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randint(0, 256, (3, h, w), generator=g, device=device).float()


class SyntheticSupportSetLoader:
    """One item (a list of length 1, batch size 1) per class:
    [{"support_set": [S x {"image", "instances"(gt_boxes, gt_classes)}], "support_set_target": LongTensor,
      "class_name": str}]"""

    def __init__(self, num_classes: int, shots: int, height: int, width: int, device="cuda", seed: int = 0,
                 shard: bool = True):
        self.ids = list(range(*inference_shard(num_classes))) if shard else list(range(num_classes))
        self.num_items = num_classes  # over ALL ranks: what the code gather sizes its per-rank block from (no count exchange)
        self.shots, self.h, self.w, self.device, self.seed = shots, height, width, device, seed

    def __len__(self):
        return len(self.ids)

    def __iter__(self):
        for c in self.ids:
            g = torch.Generator().manual_seed(self.seed * 7919 + c)
            recs = []
            for s in range(self.shots):
                x0 = torch.rand(1, generator=g).item() * 0.5 * self.w
                y0 = torch.rand(1, generator=g).item() * 0.5 * self.h
                m = 0.5 * min(self.h, self.w)
                bw = 32 + torch.rand(1, generator=g).item() * (m - 32)
                bh = 32 + torch.rand(1, generator=g).item() * (m - 32)
                inst = Instances((self.h, self.w))
                inst.gt_boxes = Boxes(torch.tensor([[x0, y0, x0 + bw, y0 + bh]]))
                inst.gt_classes = torch.tensor([c])
                recs.append({"image": _image(self.h, self.w, self.seed * 104729 + c * 131 + s, self.device),
                             "instances": inst, "height": self.h, "width": self.w})
            yield [{"support_set": recs, "support_set_target": torch.tensor(c), "class_name": f"class_{c}"}]


class SyntheticQueryLoader:
    """Batches of {"image": (3,H,W) float BGR 0-255, "height", "width", "image_id"} dicts."""

    def __init__(self, num_images: int, height: int, width: int, batch_size: int = 1, device="cuda", seed: int = 1,
                 shard: bool = True):
        self.ids = list(range(*inference_shard(num_images))) if shard else list(range(num_images))
        self.num_items = num_images
        self.h, self.w, self.bs, self.device, self.seed = height, width, batch_size, device, seed

    def __len__(self):
        return (len(self.ids) + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(0, len(self.ids), self.bs):
            yield [{"image": _image(self.h, self.w, self.seed * 15485863 + j, self.device), "height": self.h,
                    "width": self.w, "image_id": j} for j in self.ids[i:i + self.bs]]


class SyntheticBaseSupportLoader:
    """Base-class "use all ground truths" items (sylph/data/data_injection/meta_lvis.py:285-306 chunking, consumed by
    inference_on_support_set_dataset_base, meta_learn_evaluation.py:118-254): a class with total_len shots arrives as
    chunks of <= chunk shots, each item carrying "len" and "total_len".  The chunk list is sharded across ranks in
    contiguous blocks (InferenceSampler), so one class can straddle two ranks."""

    def __init__(self, shots_per_class: List[int], height: int, width: int, chunk: int = 10, device="cuda", seed: int = 0,
                 shard: bool = True):
        self.items = []
        for c, total in enumerate(shots_per_class):
            for s0 in range(0, total, chunk):
                self.items.append((c, s0, min(chunk, total - s0), total))
        self.num_items = len(self.items)
        lo, hi = inference_shard(len(self.items)) if shard else (0, len(self.items))
        self.items = self.items[lo:hi]
        self.h, self.w, self.device, self.seed = height, width, device, seed

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        for c, s0, n, total in self.items:
            recs = []
            for s in range(s0, s0 + n):
                g = torch.Generator().manual_seed(self.seed * 7919 + c * 1009 + s)
                x0 = torch.rand(1, generator=g).item() * 0.5 * self.w
                y0 = torch.rand(1, generator=g).item() * 0.5 * self.h
                m = 0.5 * min(self.h, self.w)
                bw = 32 + torch.rand(1, generator=g).item() * (m - 32)
                bh = 32 + torch.rand(1, generator=g).item() * (m - 32)
                inst = Instances((self.h, self.w))
                inst.gt_boxes = Boxes(torch.tensor([[x0, y0, x0 + bw, y0 + bh]]))
                inst.gt_classes = torch.tensor([c])
                recs.append({"image": _image(self.h, self.w, self.seed * 104729 + c * 131 + s, self.device),
                             "instances": inst, "height": self.h, "width": self.w})
            yield [{"support_set": recs, "support_set_target": torch.tensor(c), "class_name": f"class_{c}", "len": n,
                    "total_len": total}]
