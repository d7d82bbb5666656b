"""CPU checks of oracle/bf16.py (the bf16-storage restatement used to pin the production kernels): it must be the fp32
oracle up to bf16 rounding, its rounding helper must be round-to-nearest-even, and its fma a single rounding."""
import numpy as np
import torch


def test_round_is_nearest_even_and_idempotent():
    from oracle import bf16 as B
    x = torch.tensor([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20, -3.1415926, 0.0])
    y = B.r(x)
    assert y.tolist()[:4] == [1.0, 1.0, 1.0 + 2.0 ** -6, 1.0 + 2.0 ** -7]  # ties go to the even mantissa
    assert torch.equal(B.r(y), y)


def test_fma_is_one_rounding():
    from oracle import bf16 as B
    a = torch.tensor([1.0 + 2.0 ** -12]); x = torch.tensor([1.0 + 2.0 ** -12]); b = torch.tensor([-1.0])
    exact = (1.0 + 2.0 ** -12) ** 2 - 1.0
    assert float(B.fma(x, a, b)) == np.float32(exact)
    assert float(x * a + b) != np.float32(exact)  # two roundings lose the 2^-24 term


def test_bf16_oracle_tracks_fp32_oracle():
    from oracle import backbone as OB, bf16 as B, head as OH
    from sylph_amd import synthetic as W
    sd = W.synthetic_state_dict(0, 50)
    q = W.synthetic_images(1, 128, 160, seed=3)
    x16, _ = B.preprocess(q)
    x32, _ = OB.preprocess(q)
    f16, f32 = B.backbone_fpn(x16, sd), OB.backbone_fpn(x32, sd)
    for a, b in zip(f16, f32):
        assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max())
        assert torch.equal(B.r(a), a)  # everything the graph stores is bf16-representable
    codes = W.synthetic_codes(5, seed=4, scale=3.0)
    for a, b in zip(B.fcos_head(f16, sd, codes), OH.fcos_head(f32, sd, codes)):
        for l in range(5):
            assert float((a[l] - b[l]).abs().max()) <= 4e-2 * max(1.0, float(b[l].abs().max()))


def test_gn_coef_matches_group_norm():
    from oracle import bf16 as B
    g = torch.Generator().manual_seed(0)
    v = torch.randn(2, 256, 9, 11, generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    cf = B.gn_coef(v, gamma, beta)
    y = v * cf[:, :, 0].reshape(2, 256, 1, 1) + cf[:, :, 1].reshape(2, 256, 1, 1)
    ref = torch.nn.functional.group_norm(v, 32, gamma, beta, eps=1e-5)
    assert float((y - ref).abs().max()) < 1e-5
