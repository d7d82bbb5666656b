#!/usr/bin/env python
"""Micro-benchmark of the fused res2 identity bottleneck (bottleneck.hip) at the production shape: B x 200 x 336, C 256, mid 64.
HIP-event time of the kernel launch alone (sylph_profile), so the layout conversions of the parity entry are not in it.
Usage (GPU box): [SYLPH_LIB_PATH=lib/variants/...so] python tools/bench_bottleneck.py [batch] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from sylph_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
g = torch.Generator().manual_seed(0)
eng = Engine(None, dtype="bf16")
x = torch.randn(B, 256, 200, 336, generator=g).cuda()
ws = [torch.randn(64, 256, 1, 1, generator=g) / 16, torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(256, 64, 1, 1, generator=g) / 8]
sc = [torch.rand(64, generator=g) + 0.5, torch.rand(64, generator=g) + 0.5, torch.rand(256, generator=g) + 0.5]
sh = [torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1]
eng.bottleneck(x, ws, sc, sh, 1)
eng.profile_enable(True)
for _ in range(iters):
    eng.bottleneck(x, ws, sc, sh, 1)
torch.cuda.synchronize()
for name, k in eng.profile_read()["kernels"].items():
    ms, fl, n = k["ms"], k["flops"], max(k["launches"], 1)
    pos = B * 200 * 336
    print(f"{os.environ.get('SYLPH_LIB_PATH', 'product')[-40:]:40s} {name:28s} {ms / n * 1e3:8.1f} us/launch  "
          f"{fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s  {pos * 1024 / (ms / n * 1e-3) / 1e12:5.2f} TB/s (x once + y once)")
