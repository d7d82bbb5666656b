#!/usr/bin/env python
"""Soak / race check of the hand-synchronised kernels (python tools/soak_determinism.py [batch] [iterations] [classes]): the same
query batch through preprocess -> backbone -> head -> decode N times must give bit-identical pyramids, head outputs and detections every time (all reductions run in a fixed order)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from bench import make_cfg, dev_images  # noqa: E402
from sylph_amd import synthetic as W  # noqa: E402
from sylph_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
WAYS = int(sys.argv[3]) if len(sys.argv) > 3 else 5  # > 32: the fused conv + scan head (candidates appended through atomics in any order)
dev = torch.device("cuda", 0)
cfg = make_cfg()
if WAYS > 32:
    cfg.MODEL.FCOS.POST_NMS_TOPK_TEST = 300
eng = Engine(cfg, dtype="bf16", device=0)
eng.load_state_dict(W.synthetic_state_dict(0, depth=50))
g = torch.Generator().manual_seed(1)
if WAYS == 5:
    cls_conv = (torch.randn(5, 256, 1, 1, generator=g) * 0.05).to(dev)
    cls_bias = torch.full((5,), -2.0, device=dev)
else:
    codes = W.synthetic_codes(WAYS, seed=3, scale=1.5)
    cls_conv, cls_bias = codes["cls_conv"].to(dev), codes["cls_bias"].to(dev)
q = dev_images(B, 800, 1333, 3, dev)
ref = None
bad = 0
for it in range(N):
    eng.preprocess(q); eng.backbone(); eng.head(cls_conv, cls_bias)
    pyr = [t.clone() for t in eng.export_pyramid()]
    dets = eng.decode()  # before export_head: with > 32 classes the export runs the unfused conv (and its GroupNorm apply in place)
    lo, rg, ct, io = eng.export_head()
    cur = pyr + [t.clone() for t in lo + rg + ct] + [d["pred_boxes"].clone() for d in dets] + [d["scores"].clone() for d in dets]
    if ref is None:
        ref = cur
        continue
    for k, (a, b) in enumerate(zip(ref, cur)):
        if a.shape != b.shape or not torch.equal(a, b):
            bad += 1
            print(f"iteration {it}: tensor {k} differs (max abs diff {(a.float() - b.float()).abs().max().item() if a.shape == b.shape else 'shape'})")
            break
print(f"B={B}, {WAYS} classes: {N} iterations, {bad} mismatching iterations, {sum(len(d['scores']) for d in dets)} detections in the last batch")
sys.exit(1 if bad else 0)
