"""Parity of the conv kernel variants that the small golden shapes do not select by themselves: the halo-tile
mode of conv_igemm (picked only when its patches waste < 50 % of a launch) and conv_hpipe_kernel.

conv_hpipe_kernel (256x256 deep-pipelined halo conv, two patches per block) is picked automatically only for
launches with >= 512 blocks, so: (1) convs large enough to select it are compared with torch (map sizes that give
ragged patches, odd patch counts, Cout 256 and 512), and (2) the head / episode parity tests are re-run in a
subprocess with SYLPH_CONV_HPIPE=2, which forces it for every eligible layer (the FCOS towers with their fused
GroupNorm statistics, FPN output convs, code-generator tower) at the small golden sizes."""
import os
import subprocess
import sys

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
    _rerun({"SYLPH_CONV_HPIPE": "2"}, "conv2d or head or episode or backbone or codegen or full or roi_encoder")


def test_parity_suite_with_halo_mode_forced():
    """SYLPH_CONV_HALO=2 selects the halo-tile mode for every eligible 3x3 stride-1 conv whatever the patch waste
    (SYLPH_CONV_HPIPE=0 keeps the 256-wide layers on it too): conv2d vs torch, head / decode / codegen goldens,
    backbone and episode vs the oracle all run through it."""
    _rerun({"SYLPH_CONV_HALO": "2", "SYLPH_CONV_HPIPE": "0"})


def test_parity_suite_with_fusions_off():
    """The unfused graph (res2 identity blocks as three launches, stem and max-pool as two, stand-alone GroupNorm applies)
    must pass the same backbone / head / episode checks as the default fused one."""
    _rerun({"SYLPH_FUSE_BOTTLENECK": "0", "SYLPH_FUSE_STEM_POOL": "0", "SYLPH_GN_FUSE": "0"},
           "stem or backbone or head or episode or c3 or full_size_prop")
