import os, sys
sys.path.insert(0, "sylph-few-shot-detection_amd")
from sylph_amd.engine import Engine
eng = Engine(None, dtype="bf16")
gn = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ms, tf = eng.bench_conv(64, 100, 168, 256, 256, 3, 1, 1, False, True, bool(gn), iters=1)
print("RESULT", ms, tf)
