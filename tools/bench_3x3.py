#!/usr/bin/env python
"""Micro-benchmark of the MFMA-bound 3x3 layers (R-50-FPN 800x1344 shapes): python tools/bench_3x3.py [batch] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from sylph_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
LAYERS = [("tower p3 +gn", 100, 168, 256, 256, 1), ("tower p3 gn-in+gn", 100, 168, 256, 256, 3), ("tower p4 +gn", 50, 84, 256, 256, 1), ("fpn.out3", 100, 168, 256, 256, 0),
          ("res4.conv2", 50, 84, 256, 256, 0), ("res5.conv2", 25, 42, 512, 512, 0)]
eng = Engine(None, dtype="bf16")
out = []
for name, H, W, ci, co, gn in LAYERS:
    ms, tf = eng.bench_conv(B, H, W, ci, co, 3, 1, 1, False, True, gn, iters=IT)
    out.append(f"{name} {ms * 1e3:7.1f}us {tf:6.0f}TF")
print(os.environ.get("SYLPH_LIB_PATH", "default").split("libsylph_")[-1], " | ".join(out))
