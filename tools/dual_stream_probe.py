#!/usr/bin/env python
"""Experiment: do two half-batches on two HIP streams fill each other's kernel tails?  (backbone + head only)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from bench import make_cfg, dev_images
from sylph_amd import synthetic as W
from sylph_amd.engine import Engine

dev = torch.device("cuda", 0)
sd = W.synthetic_state_dict(0, depth=50)
N = 5
cls_conv = torch.randn(N, 256, 1, 1, device=dev) * 0.05
cls_bias = torch.zeros(N, device=dev) - 4.0


def run(nstreams, B, iters=8):
    engs = [Engine(make_cfg(), dtype="bf16", device=0) for _ in range(nstreams)]
    for e in engs: e.load_state_dict(sd)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    qs = [dev_images(B, 800, 1333, 7 + i, dev) for i in range(nstreams)]
    def step():
        for e, s, q in zip(engs, streams, qs):
            with torch.cuda.stream(s):
                e.preprocess(q); e.backbone(); e.head(cls_conv, cls_bias)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"streams={nstreams} B/stream={B}: {nstreams * B * iters / dt:8.1f} img/s (no decode)", flush=True)
    del engs

import sys as _s
for spec in (_s.argv[1:] or ["1x64", "2x32", "2x64", "3x64", "1x64"]):
    n, b = spec.split("x"); run(int(n), int(b))
