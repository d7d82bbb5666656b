"""Third-party pin of the ResNet half of the oracle: detectron2's ResNet is not under /root/reference, so
oracle/backbone.py restates it from its published definition.  Hugging Face `transformers.ResNetModel` is an INDEPENDENT
implementation of the same network (7x7 s2 stem + BatchNorm + ReLU, 3x3 s2 p1 max-pool, bottleneck blocks, projection shortcuts) and
its `downsample_in_bottleneck=True` is detectron2's STRIDE_IN_1X1 (the MSRA layout Sylph's yamls use): with the oracle's weights copied
into it -- FrozenBN = eval-mode BatchNorm2d, eps 1e-5 in both -- stage outputs res2..res5 must agree to fp32 rounding.  R-50 and
R-101, square and ragged (odd-sized) inputs.  CPU only; skipped when transformers is absent."""
import numpy as np
import pytest
import torch

transformers = pytest.importorskip("transformers")

from oracle import backbone as OB  # noqa: E402
from sylph_amd import synthetic as W  # noqa: E402

BLOCKS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}


def _hf_resnet(sd, depth):
    from transformers import ResNetConfig, ResNetModel
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=BLOCKS[depth],
                       layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=True)
    m = ResNetModel(cfg).eval()
    hf = m.state_dict()
    p = "backbone.bottom_up"

    def put(dst, src):  # conv weight + the four BatchNorm tensors
        hf[f"{dst}.convolution.weight"].copy_(sd[f"{src}.weight"])
        for a, b in (("weight", "weight"), ("bias", "bias"), ("running_mean", "running_mean"), ("running_var", "running_var")):
            hf[f"{dst}.normalization.{a}"].copy_(sd[f"{src}.norm.{b}"])

    put("embedder.embedder", f"{p}.stem.conv1")
    for s, n in enumerate(BLOCKS[depth]):
        for b in range(n):
            for k in range(3):
                put(f"encoder.stages.{s}.layers.{b}.layer.{k}", f"{p}.res{s + 2}.{b}.conv{k + 1}")
            if b == 0:
                put(f"encoder.stages.{s}.layers.0.shortcut", f"{p}.res{s + 2}.0.shortcut")
    m.load_state_dict(hf)
    return m


@pytest.mark.parametrize("depth,h,w", [(50, 64, 96), (50, 75, 118), (101, 64, 64)])
def test_oracle_resnet_matches_huggingface_resnet(depth, h, w):
    sd = W.backbone_state_dict(0, depth=depth)
    g = torch.Generator().manual_seed(depth + h)
    x = torch.randn(2, 3, h, w, generator=g)
    with torch.no_grad():
        ref = _hf_resnet(sd, depth)(x, output_hidden_states=True).hidden_states  # (stem + pool, res2, res3, res4, res5)
        got = OB.resnet(x, sd, depth)
    for k, name in enumerate(("res2", "res3", "res4", "res5")):
        a, b = got[name], ref[k + 1]
        assert tuple(a.shape) == tuple(b.shape), (name, a.shape, b.shape)
        err = float((a - b).abs().max())
        scale = max(1.0, float(b.abs().max()))
        assert err <= 2e-5 * scale, f"{name}: max err {err} at scale {scale}"
    assert np.isfinite(ref[-1].numpy()).all()


def test_oracle_nms_matches_greedy_nms_on_huggingface_box_iou():
    """Class-aware greedy NMS of the oracle (torchvision nms arithmetic restated, oracle/decode.py) against a plain greedy loop over
    Hugging Face's `box_iou` (transformers/loss/loss_for_object_detection.py: the pairwise-IoU arithmetic of torchvision.ops.box_iou,
    an independent copy): identical keep lists on random overlapping boxes, many per class, incl. exact score ties."""
    from transformers.loss.loss_for_object_detection import box_iou
    from oracle import decode as OD
    g = torch.Generator().manual_seed(3)
    n = 600
    xy = torch.rand(n, 2, generator=g) * 200
    wh = torch.rand(n, 2, generator=g) * 80 + 4
    boxes = torch.cat([xy, xy + wh], dim=1)
    boxes[100:140] = boxes[:40] + torch.rand(40, 4, generator=g) * 3  # near-duplicates: plenty of IoU > 0.6 pairs
    scores = torch.rand(n, generator=g)
    scores[200:220] = scores[100:120]  # exact ties
    classes = torch.randint(0, 4, (n,), generator=g)
    keep = OD.nms_per_class(boxes.numpy(), scores.numpy(), classes.numpy(), 0.6)
    iou, _ = box_iou(boxes, boxes)
    order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
    alive, want = [True] * n, []
    for a, i in enumerate(order):
        if not alive[i]:
            continue
        want.append(i)
        for j in order[a + 1:]:
            if alive[j] and int(classes[j]) == int(classes[i]) and float(iou[i, j]) > 0.6:
                alive[j] = False
    assert len(want) < n - 30, "the case must suppress a fair number of boxes"
    assert keep.tolist() == want
