#!/usr/bin/env python
"""GPU check of the deep-pipelined conv kernel against torch (run with SYLPH_CONV_PIPE=2 to force it on small shapes)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from sylph_amd.engine import Engine  # noqa: E402

eng = Engine(None, dtype="bf16")
torch.manual_seed(0)
bad = 0
for (B, H, W, ci, co, k, s, p) in [(2, 40, 56, 64, 256, 3, 1, 1), (1, 17, 23, 256, 512, 3, 1, 1), (2, 33, 47, 128, 256, 3, 2, 1),
                                   (3, 64, 80, 256, 256, 3, 1, 1)]:
    for rep in range(3):
        x = (torch.randn(B, ci, H, W) * 0.5).bfloat16().float()
        w = (torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5).bfloat16().float()
        sc, sh = torch.rand(co) + 0.5, torch.randn(co) * 0.1
        y = eng.conv2d(x, w, sc, sh, stride=s, pad=p, relu=True).cpu()
        ref = F.relu(F.conv2d(x, w, stride=s, padding=p) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        err = (y - ref).abs().max().item()
        tol = 2e-2 * max(1.0, ref.abs().max().item())
        ok = err < tol
        bad += (not ok)
        print(f"B{B} {H}x{W} {ci}->{co} k{k} s{s}: max err {err:.4g} (ref max {ref.abs().max().item():.3g}) {'ok' if ok else 'FAIL'}")
print("FAILURES" if bad else "ALL OK", bad)
