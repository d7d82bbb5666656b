#!/usr/bin/env python
"""Synchronous query steps (decode read-back per step, the protocol of the reference's loop) at batch 1 / 2 / 4 / 8 / 16, 800x1333, R-50
5-way bf16 -> img/s per batch size.  A/B the small-batch restructuring with SYLPH_SPLIT_K=0 / SYLPH_HEAD_STREAMS=0.
Usage (GPU box): python tools/sweep_small_batches.py"""
import sys, time, torch
sys.path.insert(0, "sylph-few-shot-detection_amd"); sys.path.insert(0, ".")
import bench
from sylph_amd import synthetic as W
from sylph_amd.engine import Engine
eng = Engine(bench.make_cfg(), dtype="bf16"); eng.load_state_dict(W.synthetic_state_dict(0, depth=50))
codes = W.synthetic_codes(5, seed=4, scale=3.0); cw, cb = codes["cls_conv"].cuda(), codes["cls_bias"].cuda()
out = []
for b in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 2, 4, 8, 16)):
    q = bench.dev_images(b, 800, 1333, 7, torch.device("cuda"))
    def step():
        eng.preprocess(q); eng.backbone(); eng.head(cw, cb); return eng.decode()
    for _ in range(5): step()
    torch.cuda.synchronize(); n = 60 if b <= 2 else 30; t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); out.append(f"B{b} {b * n / (time.perf_counter() - t):.1f}")
print("  ".join(out))
