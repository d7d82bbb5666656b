"""sylph_amd: MI355X-native inference path of Sylph's MetaOneStageDetector (ResNet-FPN backbone, FCOS
towers, class-conditional classifier, hypernetwork code generator, decode + NMS) behind the
reference's runner / model / predictor API.  All tensor compute is in libsylph_hip.so
(hand-written HIP for gfx950, C ABI in include/sylph_hip.h); there is no CPU fallback."""
from . import config  # noqa: F401
from .config import CfgNode, get_default_cfg  # noqa: F401

__all__ = ["config", "CfgNode", "get_default_cfg"]
