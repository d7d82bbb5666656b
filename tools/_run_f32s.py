import sys, time, torch
sys.path.insert(0, "sylph-few-shot-detection_amd"); sys.path.insert(0, ".")
import bench
from sylph_amd import synthetic as W
from sylph_amd.engine import Engine
mode = sys.argv[1] if len(sys.argv) > 1 else "f32s"
eng = Engine(bench.make_cfg(), dtype=mode); eng.load_state_dict(W.synthetic_state_dict(0, depth=50))
codes = W.synthetic_codes(5, seed=4, scale=3.0); cw, cb = codes["cls_conv"].cuda(), codes["cls_bias"].cuda()
out = []
for b in (8, 16, 32, 48, 64):
    q = bench.dev_images(b, 800, 1333, 7, torch.device("cuda"))
    def step():
        eng.preprocess(q); eng.backbone(); eng.head(cw, cb); return eng.decode()
    for _ in range(2): step()
    torch.cuda.synchronize(); n = 4; t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); out.append(f"B{b} {b * n / (time.perf_counter() - t):.1f}")
print(mode, "  ".join(out))
