"""CPU oracle for the Sylph MetaOneStageDetector inference path.

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 torch-CPU restatement of the
reference algorithm (facebookresearch/sylph-few-shot-detection) for the hot path listed in
SURVEY.md section 8.  It exists to check the HIP path; it is never the thing shipped or measured
as the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package (``sylph_amd``) never imports ``oracle``.

Parity pinning: the reference's own tests hold no numeric golden for this path.
The Sylph-owned arithmetic (FCOS head, class-conditional conv, decode, code generator, code
normalisation, class-code formatting/reduction) is pinned by golden vectors generated from the
reference itself, imported in the build container with a test-only shim
(``tests/golden/gen_goldens.py``); the fixtures are committed under ``tests/golden/``.
The third-party arithmetic that is NOT under /root/reference (detectron2 ResNet/FPN/FrozenBN/
ImageList/ROIPooler, AdelaiDet LastLevelP6P7/compute_locations/ml_nms/detector_postprocess,
torchvision roi_align/nms; all pinned only to a branch in requirements.txt:1-7) is restated
from the published operator definitions: for those pieces parity is UNPINNED by the
reference (hand-derived known-answer cases in tests/ only).

Every function cites the reference file:line (paths relative to /root/reference) it follows.
"""

from . import backbone, codegen, decode, episode, head, roi_align, roi_encoder  # noqa: F401
