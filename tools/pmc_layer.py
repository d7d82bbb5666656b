#!/usr/bin/env python
"""Run ONE conv layer shape in a loop (for rocprofv3 --pmc passes).  Usage: pmc_layer.py <name-substring> [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sylph_amd.engine import Engine  # noqa: E402

LAYERS = {
    "res2.conv3": (200, 336, 64, 256, 1, 1, 0, 1, 0),
    "res2.conv1": (200, 336, 256, 64, 1, 1, 0, 0, 0),
    "res3.conv3": (100, 168, 128, 512, 1, 1, 0, 1, 0),
    "tower": (100, 168, 256, 256, 3, 1, 1, 0, 1),
    "tower.gnin": (100, 168, 256, 256, 3, 1, 1, 0, 3),  # GroupNorm+ReLU of the previous layer applied to the halo in LDS
    "fpn.out3": (100, 168, 256, 256, 3, 1, 1, 0, 0),
    "res4.conv2": (50, 84, 256, 256, 3, 1, 1, 0, 0),
    "res3.conv2": (100, 168, 128, 128, 3, 1, 1, 0, 0),
    "res4.conv1": (50, 84, 1024, 256, 1, 1, 0, 0, 0),   # conv_pw_kernel<128, 256>
    "res5.conv1": (25, 42, 2048, 512, 1, 1, 0, 0, 0),   # conv_pw_kernel<128, 256>
    "res4.conv3": (50, 84, 256, 1024, 1, 1, 0, 1, 0),   # conv_igemm (residual)
}
name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, W, ci, co, k, s, p, res, gn = LAYERS[name]
eng = Engine(None, dtype="bf16")
ms, tf = eng.bench_conv(B, H, W, ci, co, k, s, p, bool(res), True, int(gn), iters=5)
print(name, ms * 1e3, "us", tf, "TF")
