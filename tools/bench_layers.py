#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv kernel on the R-50-FPN 800x1344 layer shapes.
Usage (GPU box): python tools/bench_layers.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
from sylph_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LAYERS = [  # name, H, W, Cin, Cout, K, stride, pad, res, gn
    ("res2.conv1 1x1 256->64", 200, 336, 256, 64, 1, 1, 0, 0, 0),
    ("res2.conv2 3x3 64->64", 200, 336, 64, 64, 3, 1, 1, 0, 0),
    ("res2.conv3 1x1 64->256 +res", 200, 336, 64, 256, 1, 1, 0, 1, 0),
    ("res3.conv1 1x1 512->128", 100, 168, 512, 128, 1, 1, 0, 0, 0),
    ("res3.conv2 3x3 128->128", 100, 168, 128, 128, 3, 1, 1, 0, 0),
    ("res3.conv3 1x1 128->512 +res", 100, 168, 128, 512, 1, 1, 0, 1, 0),
    ("res4.conv1 1x1 1024->256", 50, 84, 1024, 256, 1, 1, 0, 0, 0),
    ("res4.conv2 3x3 256->256", 50, 84, 256, 256, 3, 1, 1, 0, 0),
    ("res4.conv3 1x1 256->1024 +res", 50, 84, 256, 1024, 1, 1, 0, 1, 0),
    ("res5.conv1 1x1 2048->512", 25, 42, 2048, 512, 1, 1, 0, 0, 0),
    ("res5.conv2 3x3 512->512", 25, 42, 512, 512, 3, 1, 1, 0, 0),
    ("res5.conv3 1x1 512->2048 +res", 25, 42, 512, 2048, 1, 1, 0, 1, 0),
    ("fpn.out3 3x3 256->256", 100, 168, 256, 256, 3, 1, 1, 0, 0),
    ("tower(p3 only) 3x3 256->256 +gn", 100, 168, 256, 256, 3, 1, 1, 0, 1),
]
eng = Engine(None, dtype="bf16")
for name, H, W, ci, co, k, s, p, res, gn in LAYERS:
    ms, tf = eng.bench_conv(B, H, W, ci, co, k, s, p, bool(res), True, bool(gn), iters=10)
    rows = B * (H // s) * (W // s)
    byts = rows * (ci + co * (2 if res else 1)) * 2
    print(f"{name:34s} {ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s  {byts / (ms * 1e-3) / 1e12:6.2f} TB/s(min traffic)")
