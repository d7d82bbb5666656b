"""Parity of the conv kernel variants that the small golden shapes do not select by themselves: the halo-tile
mode of conv_igemm (picked only when its patches waste < 50 % of a launch) and conv_hpipe_kernel.

The forced-variant reruns select the bf16 tests of tests/test_hip_parity.py that can reach the forced kernel (the variants exist in
bf16 mode only; the fp32 / split-bf16 checks of that file are untouched by the knobs and run once, in the main suite).

conv_hpipe_kernel (256x256 deep-pipelined halo conv, two patches per block) is picked automatically only for
launches with >= 512 blocks, so: (1) convs large enough to select it are compared with torch (map sizes that give
ragged patches, odd patch counts, Cout 256 and 512), and (2) the head / episode parity tests are re-run in a
subprocess with SYLPH_CONV_HPIPE=2, which forces it for every eligible layer (the FCOS towers with their fused
GroupNorm statistics, FPN output convs, code-generator tower) at the small golden sizes."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [(2, 64, 256, 256, 256, 1), (2, 128, 250, 270, 256, 1), (3, 64, 203, 171, 512, 1),
                                  (3, 64, 300, 310, 512, 2)])
def test_large_conv_selects_hpipe_kernel_and_matches_torch(case):
    from sylph_amd.engine import Engine
    B, C, H, W, Cout, stride = case
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = (torch.randn(B, C, H, W, generator=g) * 0.5).bfloat16().float()
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (C * 9) ** 0.5).bfloat16().float()
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    eng = Engine(None, dtype="bf16")
    y = eng.conv2d(x, w, scale, shift, stride, 1, True)
    ref = F.relu(F.conv2d(x.cuda(), w.cuda(), None, stride, 1) * scale.cuda().view(1, -1, 1, 1) + shift.cuda().view(1, -1, 1, 1))
    err = (y - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), f"max err {err}"


def _rerun(env_extra, k=None):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_parity.py"), "-m", "gpu", "-q", "-x"]
    if k:
        cmd += ["-k", k]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_parity_suite_with_hpipe_kernel_forced():
    _rerun({"SYLPH_CONV_HPIPE": "2"}, "bf16 and (conv2d or head or episode or backbone_fpn or codegen_tower or c3_full_size or full_size_prop)")


def test_parity_suite_with_halo_mode_forced():
    """SYLPH_CONV_HALO=2 selects the halo-tile mode for every eligible 3x3 stride-1 conv whatever the patch waste
    (SYLPH_CONV_HPIPE=0 keeps the 256-wide layers on it too): conv2d vs torch, head / decode / codegen goldens,
    backbone and episode vs the oracle all run through it."""
    _rerun({"SYLPH_CONV_HALO": "2", "SYLPH_CONV_HPIPE": "0"}, "bf16 and (conv2d or head or episode or backbone_fpn or codegen or c3_full_size or full_size_prop)")


def test_parity_suite_with_streaming_residual_conv_everywhere():
    """SYLPH_CONV_SPW=2 routes EVERY layer conv_spw_kernel can run through it whatever the launch size -- bottleneck conv3 with a
    same-geometry residual (K 128 / 256 / 512), and since round 6 conv3 + projection shortcut of the first res3 block (two inputs, the
    second strided), the stride-2 conv1 of the first res4 block and FPN lateral3 with its nearest-2x top-down add -- (weights
    in registers, A ring, residual and result streamed through the LDS tile buffer by the streaming waves) whatever the launch size --
    ragged batches, partial last tiles, launches of a few tiles: conv2d vs torch, backbone / episode / full-size checks against the
    oracle, and the ulp-level block tests."""
    env = {"SYLPH_CONV_SPW": "2"}
    _rerun(env, "bf16 and (conv2d or backbone_fpn or c3_full_size or full_size_prop)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_bf16_pinned_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "bottleneck or lateral"], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_parity_suite_with_register_weight_conv2_everywhere():
    """SYLPH_CONV_RW3=2 routes every stride-1 128 -> 128 3x3 bottleneck conv2 whose patch shape fits (conv_rw3_patch_ok) through
    conv_rw3_kernel whatever the launch size -- launches of a few patches, fewer patches than CUs, one-patch images: backbone / episode /
    full-size checks against the oracle and the ulp-level block tests."""
    env = {"SYLPH_CONV_RW3": "2"}
    _rerun(env, "bf16 and (backbone_fpn or c3_full_size or full_size_prop)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_bf16_pinned_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "bottleneck"], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_parity_suite_with_pointwise_kernel_everywhere():
    """SYLPH_CONV_PW=2 routes EVERY eligible bf16 1x1 layer through conv_pw_kernel whatever the launch size -- also the layers the
    default policy leaves on conv_igemm (same-geometry residual: the RES = 1 instantiations; N = 128 identity conv1): conv2d vs
    torch, backbone / episode / full-size checks, and the ulp-level block tests.  (The rejected 128x128 / 256x256 tile variants
    exist only in -DSYLPH_ABLATE builds, tools/build_variant.sh.)"""
    env = {"SYLPH_CONV_PW": "2"}
    _rerun(env, "bf16 and (conv2d or backbone_fpn or c3_full_size or full_size_prop)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_bf16_pinned_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "bottleneck"], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_parity_suite_with_small_tiles_on_three_stages_everywhere():
    """Round 6's small-launch forms on EVERY launch that takes conv_igemm's 64-row tiles, whatever its size: 64 x 64 tiles instead of
    64 x 128 (SYLPH_CONV_BN64_MAX) walking K through three LDS stages (SYLPH_CONV_NBUF3_MAX), with the pointwise / streaming kernels
    off so that the 1x1 layers come this way too, K never split (SYLPH_SPLIT_K=0): conv2d vs torch, backbone / head / episode vs the
    oracle, the full-size checks, and the ulp-level block tests."""
    env = {"SYLPH_CONV_BN64_MAX": "100000000", "SYLPH_CONV_NBUF3_MAX": "100000000", "SYLPH_CONV_PW": "0", "SYLPH_CONV_SPW": "0", "SYLPH_SPLIT_K": "0"}
    _rerun(env, "bf16 and (conv2d or backbone_fpn or head or episode or c3_full_size or full_size_prop)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_bf16_pinned_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "bottleneck"], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_parity_suite_with_split_k_and_two_streams_forced():
    """SYLPH_SPLIT_K=2 splits EVERY eligible bf16 conv_igemm launch along K (fp32 partial planes + splitk_finish_kernel), whatever its
    tile count or depth -- 1x1 and 3x3, with and without a same-geometry residual, strided -- and SYLPH_HEAD_STREAMS=2 always runs the
    bbox tower on the second stream: conv2d vs torch, backbone / episode vs the oracle, the full-size and the ulp-level block tests."""
    env = {"SYLPH_SPLIT_K": "2", "SYLPH_HEAD_STREAMS": "2"}
    _rerun(env, "bf16 and (conv2d or backbone_fpn or c3_full_size or full_size_prop or head)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_bf16_pinned_gpu.py"), "-m", "gpu", "-q", "-x", "-k",
                        "bottleneck"], env=dict(os.environ, **env), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_small_batch_restructuring_is_result_neutral():
    """Batch 1 (split-K where it pays + two streams, the default) against the same image inside a batch of 8 (neither): identical
    detections -- same (level, location, class) triples -- and pyramids equal to bf16 rounding of the re-ordered fp32 sums."""
    child = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from sylph_amd import synthetic as W
from test_hip_parity import _engine, _cfg
eng = _engine("bf16", _cfg())
eng.load_state_dict(W.synthetic_state_dict(0, depth=50))
imgs = W.synthetic_images(1, 512, 672, seed=17)
codes = W.synthetic_codes(5, seed=4, scale=3.0)
eng.preprocess(imgs); eng.backbone(); eng.head(codes["cls_conv"], codes["cls_bias"])
d = eng.decode()[0]
np.savez(sys.argv[3], cand=d["cand_index"].cpu().numpy(), scores=d["scores"].cpu().numpy(), *[t.float().cpu().numpy() for t in eng.export_pyramid()])
"""
    import tempfile
    outs = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env_extra in (("default", {}), ("plain", {"SYLPH_SPLIT_K": "0", "SYLPH_HEAD_STREAMS": "0"})):
            path = os.path.join(td, f"{name}.npz")
            r = subprocess.run([sys.executable, "-c", child, os.path.join(ROOT, "sylph-few-shot-detection_amd"), os.path.join(ROOT, "tests"), path],
                               env=dict(os.environ, **env_extra), cwd=ROOT, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            z = np.load(path)
            outs[name] = {k: z[k] for k in z.files}
    a, b = outs["default"], outs["plain"]
    for k in [k for k in a if k.startswith("arr_")]:
        cos = float((a[k] * b[k]).sum() / (np.linalg.norm(a[k]) * np.linalg.norm(b[k]) + 1e-30))
        err = np.abs(a[k] - b[k]).max() / max(1.0, np.abs(b[k]).max())
        assert cos > 0.9999 and err < 0.03, f"{k}: cosine {cos}, max rel err {err}"
    common = np.intersect1d(a["cand"], b["cand"]).size
    assert a["cand"].size >= 50 and common >= 0.9 * max(a["cand"].size, b["cand"].size), (a["cand"].size, b["cand"].size, common)


def test_parity_suite_with_fusions_off():
    """The unfused graph (res2 identity blocks as three launches, stem and max-pool as two, stand-alone GroupNorm applies)
    must pass the same backbone / head / episode checks as the default fused one."""
    _rerun({"SYLPH_FUSE_BOTTLENECK": "0", "SYLPH_FUSE_STEM_POOL": "0", "SYLPH_GN_FUSE": "0", "SYLPH_FUSE_GN_LOGITS": "0"},
           "bf16 and (stem or backbone_fpn or head or episode or c3_full_size or full_size_prop)")


_PYRAMID_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from sylph_amd import synthetic as W
from test_hip_parity import _engine, _cfg
B, H, Wd = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
eng = _engine("bf16", _cfg())
eng.load_state_dict(W.synthetic_state_dict(0, depth=50))
eng.preprocess(W.synthetic_images(B, H, Wd, seed=11))
eng.backbone()
np.savez(sys.argv[3], *[t.float().cpu().numpy() for t in eng.export_pyramid()])
"""


@pytest.mark.parametrize("B,H,W", [(1, 200, 232), (3, 72, 328), (2, 264, 136), (2, 40, 56)])
def test_fused_backbone_kernels_match_unfused_graph_on_ragged_maps(tmp_path, B, H, W):
    """bottleneck64(_p)_kernel and stem_pool_kernel against the three-launch / two-launch graph they replace, on maps whose
    patches are ragged in both directions and whose tile count is not a multiple of the 8-block walk (res2 maps 50x58,
    18x82, 66x34).  Both builds round to bf16 at the same points, so the pyramids agree to bf16 noise."""
    outs = {}
    for name, env_extra in (("fused", {}), ("unfused", {"SYLPH_FUSE_BOTTLENECK": "0", "SYLPH_FUSE_STEM_POOL": "0"})):
        path = str(tmp_path / f"{name}.npz")
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", _PYRAMID_CHILD, os.path.join(ROOT, "sylph-few-shot-detection_amd"),
                            os.path.join(ROOT, "tests"), path, str(B), str(H), str(W)], env=env, cwd=ROOT, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        z = np.load(path)
        outs[name] = [z[k] for k in z.files]
    for lvl, (a, b) in enumerate(zip(outs["fused"], outs["unfused"])):
        assert a.shape == b.shape and np.isfinite(a).all()
        cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        err = np.abs(a - b).max() / max(1.0, np.abs(b).max())
        assert cos > 0.9995 and err < 0.05, f"level {lvl}: cosine {cos}, max rel err {err}"


@pytest.mark.gpu
def test_three_stage_ring_of_the_small_tiles_is_bit_identical(tmp_path):
    """conv_igemm's 64-row tiles walk K through three LDS stages in launches of at most 400 tiles (round 6: two slices in flight where
    nothing else hides a slice's round trip) and through two otherwise (SYLPH_CONV_NBUF3_MAX=0: everywhere).  Same slices, same order,
    same MFMAs: the batch-1 pyramid (res3..res5, FPN: 264- / 144- / 96-tile launches) must be bit-identical."""
    outs = {}
    for name, env_extra in (("three", {}), ("two", {"SYLPH_CONV_NBUF3_MAX": "0"})):
        path = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, "-c", _PYRAMID_CHILD, os.path.join(ROOT, "sylph-few-shot-detection_amd"), os.path.join(ROOT, "tests"), path,
                            "1", "800", "1333"], env=dict(os.environ, **env_extra), cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        z = np.load(path)
        outs[name] = [z[k] for k in z.files]
    assert len(outs["three"]) == 5
    for lvl, (a, b) in enumerate(zip(outs["three"], outs["two"])):
        assert np.isfinite(a).all() and np.array_equal(a, b), f"level {lvl}: {np.abs(a - b).max()}"
