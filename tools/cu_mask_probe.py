#!/usr/bin/env python
"""Experiment: two engine contexts on two CU-masked HIP streams (hipExtStreamCreateWithCUMask), each owning a disjoint
share of every XCD's CUs, so that one context's HBM-bound backbone layers overlap the other's MFMA-bound tower layers without
the 256x256-tile kernels of the two alternating on a CU.  Usage (GPU box): python tools/cu_mask_probe.py [B]"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from bench import make_cfg, dev_images  # noqa: E402
from sylph_amd import synthetic as W  # noqa: E402
from sylph_amd.engine import Engine  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
torch.cuda.init(); torch.zeros(1, device=dev)
sd = W.synthetic_state_dict(0, depth=50)
N = 5
cls_conv = torch.randn(N, 256, 1, 1, device=dev) * 0.05
cls_bias = torch.zeros(N, device=dev) - 4.0


def masked_stream(pred):
    """pred(i) -> bool for mask bit i (256 CUs)."""
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(tag, streams, B, iters=8, decode=False):
    n = len(streams)
    engs = [Engine(make_cfg(), dtype="bf16", device=0) for _ in range(n)]
    for e in engs: e.load_state_dict(sd)
    qs = [dev_images(B, 800, 1333, 7 + i, dev) for i in range(n)]

    def step():
        pend = []
        for e, s, q in zip(engs, streams, qs):
            with torch.cuda.stream(s):
                e.preprocess(q); e.backbone(); e.head(cls_conv, cls_bias)
                if decode: pend.append((e, s, e.decode_launch()))
        for e, s, p in pend:
            with torch.cuda.stream(s):
                e.decode_fetch(p)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{tag:44s} streams={n} B/stream={B}: {n * B * iters / dt:8.1f} img/s", flush=True)
    del engs


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
run("one stream, all CUs", [torch.cuda.Stream()], B)
run("two streams, all CUs", [torch.cuda.Stream(), torch.cuda.Stream()], B)
half = lambda i: (i // 8) % 2 == 0
run("two streams, complementary halves of each XCD", [masked_stream(half), masked_stream(lambda i: not half(i))], B)
q3 = lambda i: (i // 8) % 4 != 3
run("two streams, 3/4 + 1/4", [masked_stream(q3), masked_stream(lambda i: not q3(i))], B)
run("two streams, halves (contiguous bit ranges)", [masked_stream(lambda i: i < 128), masked_stream(lambda i: i >= 128)], B)
run("one masked stream, half the CUs", [masked_stream(half)], B)
run("one stream, all CUs", [torch.cuda.Stream()], B)
