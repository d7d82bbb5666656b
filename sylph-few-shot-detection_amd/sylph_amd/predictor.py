"""SylphPredictor: single-image few-shot predictor with the reference's constructor and call surface
(sylph/predictor.py:38-298).  Differences forced by scope: dataset catalogs are not available, so the
class list of each split comes from ``test_dataset_names[split]`` given as
{"name": <dataset name>, "thing_classes": [...]} (or a plain dataset name, in which case every
<class>.pth found in the class-code directory is used, ordered by its stored support_set_target)."""
import glob
import logging
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .evaluation import format_class_codes_shared
from .runner import create_cfg, create_runner

logger = logging.getLogger(__name__)


def resize_shortest_edge_shape(h: int, w: int, size: int, max_size: int) -> Tuple[int, int]:
    """detectron2 ResizeShortestEdge.get_output_shape (sylph/predictor.py:117-120 builds it with
    [MIN_SIZE_TEST, MIN_SIZE_TEST], MAX_SIZE_TEST)."""
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def resize_image(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """Host resize of the fallback path (fused_preprocess=False): detectron2 ResizeTransform.apply_image, i.e.
    PIL.Image.BILINEAR for uint8 images and, for any other dtype, torch F.interpolate(mode="bilinear",
    align_corners=False) WITHOUT antialiasing (detectron2 does exactly that for non-uint8 input)."""
    if img.shape[0] == new_h and img.shape[1] == new_w:
        return img
    from PIL import Image
    if img.dtype == np.uint8:
        return np.asarray(Image.fromarray(img).resize((new_w, new_h), Image.BILINEAR))
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, (new_h, new_w), mode="bilinear", align_corners=False)
    return t[0].permute(1, 2, 0).numpy()


class SylphPredictor:
    def __init__(self, config_file: str, weight_path: str, class_code_path: str,
                 runner_name: str = "sylph.runner.MetaFCOSRunner", test_dataset_names: Dict = None,
                 dtype: Optional[str] = None, fused_preprocess: bool = True):
        logger.info("SylphPredictor initializing...")
        runner = create_runner(runner_name)
        self.cfg = create_cfg(runner.get_default_cfg(), config_file, None).clone()
        assert self.cfg.MODEL.META_LEARN.EPISODIC_LEARNING, "This is not few-shot model"
        self.cfg.MODEL.WEIGHTS = weight_path
        if not torch.cuda.is_available():
            raise RuntimeError("SylphPredictor needs a ROCm GPU: the MI355X path has no CPU fallback")
        self.model = runner.build_model(self.cfg, dtype=dtype)
        self.model.eval()
        if weight_path and not self.model._weights_loaded:  # build_model already loads an existing MODEL.WEIGHTS
            self.model.load_checkpoint(weight_path)
        self.class_code_path = class_code_path
        self.metadatas, self.class_codes = {}, {}
        assert test_dataset_names is not None, "No test data"
        assert "all" in test_dataset_names, "split 'all' is not in the test dataset names"
        for split, ds in test_dataset_names.items():
            name = ds["name"] if isinstance(ds, dict) else ds
            classes = ds.get("thing_classes") if isinstance(ds, dict) else None
            self.metadatas[split] = {"name": name, "thing_classes": classes}
            if split == "all":
                self.class_codes[split] = self._get_datasets_class_codes(self.metadatas[split], name)
        self.min_size = int(self.cfg.INPUT.MIN_SIZE_TEST)
        self.max_size = int(self.cfg.INPUT.MAX_SIZE_TEST)
        self.fused_preprocess = fused_preprocess
        self._pinned = {}  # (h, w) -> pinned uint8 staging buffer for the asynchronous H2D copy
        self.input_format = self.cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR"], self.input_format
        logger.info("SylphPredictor done initialization")

    def _get_datasets_class_codes(self, metadata: Dict, dataset_name: str):
        """sylph/predictor.py:167-187: <class_code_path>/<dataset_name>/0/<class_name>.pth."""
        code_path = os.path.join(self.class_code_path, dataset_name, "0")
        classes = metadata.get("thing_classes")
        class_codes = []
        if classes is None:
            files = sorted(glob.glob(os.path.join(code_path, "*.pth")))
            if not files:
                raise ValueError(f"no class code files under {code_path}")
            class_codes = [torch.load(f, map_location="cpu", weights_only=False) for f in files]
            class_codes.sort(key=lambda c: int(c["support_set_target"]))
            metadata["thing_classes"] = [c["class_name"] for c in class_codes]
        else:
            for class_name in classes:
                f = os.path.join(code_path, f"{class_name}.pth")
                if not os.path.exists(f):
                    raise ValueError(f"{f} is missing")
                class_codes.append(torch.load(f, map_location="cpu", weights_only=False))
        logger.info(f"Got {len(class_codes)} class codes for prediction.")
        codes = format_class_codes_shared(class_codes, device=self.model.device)
        assert "cls_conv" in codes, "conv is not in class_codes"
        return codes

    def _preprocess(self, original_image: np.ndarray):
        height, width = original_image.shape[:2]
        nh, nw = resize_shortest_edge_shape(height, width, self.min_size, self.max_size)
        if self.fused_preprocess and original_image.dtype == np.uint8:
            # sylph/predictor.py:259-269 on the GPU: the uint8 original goes to the device through a pinned staging
            # buffer; resize (Pillow-exact), RGB->BGR, normalisation and padding are one kernel in front of the backbone
            buf = self._pinned.get((height, width))
            if buf is None:
                if len(self._pinned) > 16:
                    self._pinned.clear()
                buf = self._pinned[(height, width)] = torch.empty(height, width, 3, dtype=torch.uint8, pin_memory=True)
            buf.copy_(torch.from_numpy(np.ascontiguousarray(original_image)))
            return {"image_u8": buf, "resize_hw": (nh, nw), "input_format": self.input_format, "height": height, "width": width}
        if self.input_format == "RGB":
            original_image = original_image[:, :, ::-1]
        image = resize_image(np.ascontiguousarray(original_image), nh, nw)
        image = torch.as_tensor(image.astype("float32").transpose(2, 0, 1))
        return {"image": image, "height": height, "width": width}

    def _call_few_shot(self, original_image: np.ndarray, class_codes: Dict[str, torch.Tensor]):
        """sylph/predictor.py:248-274: original_image (H, W, C) BGR -> {"instances": Instances}."""
        with torch.no_grad():
            inputs = self._preprocess(original_image)
            return self.model([inputs], class_code=class_codes, run_type="meta_learn_test_instance")[0]

    def inference_on_split(self, original_image: np.ndarray, split: str = "all"):
        return self._call_few_shot(original_image, self.class_codes[split])

    def __call__(self, original_image):
        """The reference's __call__ runs the base detector (run_type None), which an episodic model
        rejects; kept for surface compatibility."""
        with torch.no_grad():
            return self.model([self._preprocess(original_image)])[0]
