#!/usr/bin/env python
"""Micro-benchmark of one identity bottleneck block at its production shape (default: res2, B x 200 x 336, C 256, mid 64: the fused
bottleneck.hip kernel; stage 3 / 4 / 5: the three conv launches of a res3 / res4 / res5 identity block).
HIP-event time of the kernel launch alone (sylph_profile), so the layout conversions of the parity entry are not in it.
Usage (GPU box): [SYLPH_LIB_PATH=lib/variants/...so] python tools/bench_bottleneck.py [batch] [iters] [stage]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from sylph_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
stage = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mid, C, Hm, Wm = 64 << (stage - 2), 256 << (stage - 2), 200 >> (stage - 2), 336 >> (stage - 2)
g = torch.Generator().manual_seed(0)
eng = Engine(None, dtype="bf16")
x = torch.randn(B, C, Hm, Wm, generator=g).cuda()
ws = [torch.randn(mid, C, 1, 1, generator=g) / C ** 0.5, torch.randn(mid, mid, 3, 3, generator=g) / (9 * mid) ** 0.5,
      torch.randn(C, mid, 1, 1, generator=g) / mid ** 0.5]
sc = [torch.rand(mid, generator=g) + 0.5, torch.rand(mid, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5]
sh = [torch.randn(mid, generator=g) * 0.1, torch.randn(mid, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1]
eng.bottleneck(x, ws, sc, sh, 1)
eng.profile_enable(True)
for _ in range(iters):
    eng.bottleneck(x, ws, sc, sh, 1)
torch.cuda.synchronize()
for name, k in eng.profile_read()["kernels"].items():
    ms, fl, n = k["ms"], k["flops"], max(k["launches"], 1)
    pos = B * Hm * Wm
    print(f"{os.environ.get('SYLPH_LIB_PATH', 'product')[-40:]:40s} {name:28s} {ms / n * 1e3:8.1f} us/launch  "
          f"{fl / (ms * 1e-3) / 1e12:7.1f} TFLOP/s  {pos * 1024 / (ms / n * 1e-3) / 1e12:5.2f} TB/s (at 1 KiB / position)")
