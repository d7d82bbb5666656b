"""The split-bf16 parity mode (dtype "f32s", SYLPH_F32S): fp32 storage as in the fp32 mode, every conv product as three bf16 MFMAs on
operands split into bf16 hi + lo parts (conv_igemm.hip MmaSplit).  It has to pass the SAME checks as the exact-fp32 mode: the fp32 tests
of tests/test_hip_parity.py are re-run with SYLPH_TEST_F32_MODE=f32s (a test-harness switch: `_engine("f32")` then builds an "f32s"
engine) -- op-level convs against torch, the reference-generated goldens (head outputs 1e-3, identical (level, location, class)
triples), backbone / episode against the oracle, and the two full-size statements (800x1333 R-50 5-way, R-101 866-way)."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_split_operand_error_is_two_to_the_minus_seventeen():
    """A K = 2304 conv in the three modes against float64: the split mode's error sits between the exact fp32 mode's and 2^-17 relative
    per product (hi + lo keeps 16 mantissa bits of each operand), three orders of magnitude below bf16 storage."""
    from sylph_amd.engine import Engine
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 256, 24, 40, generator=g)
    w = torch.randn(256, 256, 3, 3, generator=g) / 48.0
    one, zero = torch.ones(256), torch.zeros(256)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    scale = ref.abs().max().item()
    err = {}
    for mode in ("f32", "f32s", "bf16"):
        y = Engine(None, dtype=mode).conv2d(x, w, one, zero, 1, 1, False).cpu().double()
        err[mode] = (y - ref).abs().max().item() / scale
    print("max error / max |ref|:", err)
    assert err["f32"] <= 2e-6
    assert err["f32s"] <= 2 ** -17
    assert err["bf16"] >= 50 * err["f32s"]


def _rerun(k):
    env = dict(os.environ, SYLPH_TEST_F32_MODE="f32s")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_parity.py"), "-m", "gpu", "-q", "-x", "-k", k]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
    return r.stdout


def test_fp32_parity_checks_pass_in_split_mode():
    out = _rerun("(f32 or golden or oracle) and not bf16")
    print(out[-400:])
