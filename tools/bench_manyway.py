#!/usr/bin/env python
"""Query throughput with many classes (BASELINE configs[3]-like head load: 866-way LVIS episode) on the R-50 / R-101 backbone.
Usage (GPU box): python tools/bench_manyway.py [depth] [batch] [ways] [code scale]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sylph-few-shot-detection_amd"))
import torch  # noqa: E402
from bench import make_cfg, dev_images  # noqa: E402
from sylph_amd import synthetic as W  # noqa: E402
from sylph_amd.engine import Engine  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 866
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
cfg = make_cfg()
cfg.MODEL.RESNETS.DEPTH = depth
cfg.MODEL.FCOS.POST_NMS_TOPK_TEST = 300
eng = Engine(cfg, dtype="bf16")
eng.load_state_dict(W.synthetic_state_dict(0, depth=depth))
dev = torch.device("cuda", 0)
q = dev_images(B, 800, 1333, 7, dev)
codes = W.synthetic_codes(N, seed=3, scale=scale)
cw, cb = codes["cls_conv"].to(dev), codes["cls_bias"].to(dev)


def step():
    eng.preprocess(q); eng.backbone(); eng.head(cw, cb)
    return eng.decode()


for _ in range(8):
    d = step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    d = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
eng.profile_enable(True); eng.profile_read()
for _ in range(n):
    step()
torch.cuda.synchronize()
p = eng.profile_read()
print(f"R-{depth} {N}-way B={B}: {B * n / dt:.1f} img/s, {dt / n * 1e3:.2f} ms/step, conv kernels {p['conv_ms'] / n:.2f} ms/step, detections {[int(x['scores'].numel()) for x in d][:4]}")
tot = 0.0
for k, v in sorted(p["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print(f"   {k:40s} {v['ms'] / n:8.3f} ms/step")
print(f"   all kernels {sum(v['ms'] for v in p['kernels'].values()) / n:8.3f} ms/step")
