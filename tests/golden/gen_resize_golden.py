"""Golden vectors of the input pipeline: REAL Pillow outputs (PIL.Image.resize(..., BILINEAR) on uint8 HWC images, the call
detectron2's ResizeTransform makes for sylph/predictor.py:259-269) on small seeded images, for up- and down-sampling ratios
including the ones ResizeShortestEdge(800, 1333) produces.  Run where Pillow is installed:

    python tests/golden/gen_resize_golden.py        # writes tests/golden/g9_resize.npz (records PIL.__version__)
"""
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [  # (h, w, new_h, new_w)
    (90, 130, 96, 139),    # ResizeShortestEdge(96, 160) of 90 x 130 (mild up-sampling)
    (48, 64, 80, 107),     # 480 x 640 -> 800 x 1067 ratio (x 1.667)
    (200, 150, 133, 100),  # down-sampling x 1.5: 5-tap antialiasing filter
    (300, 211, 97, 68),    # down-sampling x 3.1: 9-tap filter
    (37, 91, 100, 246),    # strong up-sampling
    (64, 64, 64, 64),      # identity
    (50, 60, 50, 90),      # horizontal pass only
]


def image(h, w, seed):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    smooth = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 255 // max(h + w - 2, 1))], -1)
    noise = rng.randint(0, 256, (h, w, 3))
    img = np.where(rng.rand(h, w, 1) < 0.5, smooth, noise)  # ramps (exercise rounding ties) mixed with noise
    img[: h // 8, : w // 8] = 255   # saturated and black blocks: clip8 at both ends
    img[-(h // 8 + 1):, -(w // 8 + 1):] = 0
    return img.astype(np.uint8)


if __name__ == "__main__":
    out = {"pil_version": np.array(PIL.__version__)}
    out["cases"] = np.array(CASES)
    for i, (h, w, nh, nw) in enumerate(CASES):
        img = image(h, w, 100 + i)
        out[f"in{i}"] = img
        out[f"out{i}"] = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    np.savez_compressed(os.path.join(HERE, "g9_resize.npz"), **out)
    print("wrote g9_resize.npz with Pillow", PIL.__version__)
