"""GPU: kernel-level parity of the HIP building blocks (through the C-ABI test hooks) against float64 torch-CPU.

Tolerances are relative to the natural scale of each result:
  f32    exact-fp32 MFMA (k-ordered fmaf chain): 2e-6 * sum|a||w| bound -> checked as 1e-5 relative
  f16x3  split-fp16, ~2^-22 operand error: 2e-5 relative (the fast erf-GELU / SiLU forms add ~1e-6)
  bf16x3 split-bf16, ~2^-17 operand error: 1e-4 relative
  bf16   8-bit mantissa operands: 2e-2 relative
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-5, "f16x3": 2e-5, "bf16x3": 1e-4, "bf16": 2e-2}
PRECS = list(TOL)


def _rel(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.fixture(scope="module")
def hb():
    from oakink2_tamf_amd import hip_backend

    hip_backend.require_gpu()
    return hip_backend


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K,act", [(300, 256, 192, 0), (128, 128, 64, 1), (1000, 384, 512, 2), (77, 128, 99, 0)])
def test_gemm(hb, prec, M, N, K, act):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref * torch.sigmoid(ref)
    elif act == 2:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2.0)))
    got = hb.test_gemm(prec, a.cuda(), w.cuda(), b.cuda(), act)
    assert _rel(got, ref) < TOL[prec]


@pytest.mark.parametrize("prec", PRECS)
def test_gemm_asymmetric_identity(hb, prec):
    """A = I against an asymmetric integer W catches a transposed accumulator layout exactly."""
    K = 128
    a = torch.eye(K)
    w = (torch.arange(256 * K).reshape(256, K) % 13 - 6).float() + (torch.arange(256).reshape(256, 1) % 5).float()
    got = hb.test_gemm(prec, a.cuda(), w.cuda(), None, 0).cpu()
    assert torch.equal(got, w.t().contiguous())


def _resid_case(M, N, K, seed, with_ln, dc=0.0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bb = torch.randn(N, generator=g) * 0.1
    u = torch.randn(M, N, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + dc * torch.randn(M, 1, generator=g)
    ga = 1 + 0.3 * torch.randn(N, generator=g) if with_ln else torch.ones(N)
    y = torch.nn.functional.layer_norm(u.double(), (N,), ga.double(), None, 1e-5) if with_ln else u.double()
    ref = (y + bb.double()) + a.double() @ w.double().t()
    return a, w, bb, ga, u, ref


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("with_ln", [True, False])
@pytest.mark.parametrize("M,N,K", [(200, 128, 128), (333, 256, 1024), (130, 512, 512), (64, 512, 2048), (512, 512, 512), (1000, 512, 1024),
                                   (3 * 208, 512, 512), (2 * 168, 512, 2048)])
def test_gemm_residual_with_deferred_layernorm(hb, prec, with_ln, M, N, K):
    """EpiResid through tamf_test_gemm_resid: the residual add of an encoder sublayer with the LayerNorm of its input applied on the way
    (reference: x + sublayer(x) behind norm1 / norm2, interaction_segment_mdm.py:63-70), and the block statistics it leaves"""
    a, w, bb, ga, u, ref = _resid_case(M, N, K, 2, with_ln, dc=1.5)
    got, st = hb.test_gemm_resid(prec, a.cuda(), w.cuda(), bb.cuda(), ga.cuda(), u.cuda(), hb.block_stats(u).cuda() if with_ln else None)
    assert _rel(got, ref) < TOL[prec]
    want = hb.block_stats(got.cpu())  # the statistics describe the row that was STORED (fp32)
    assert torch.allclose(st.cpu()[..., 0], want[..., 0], rtol=1e-5, atol=1e-4)
    assert torch.allclose(st.cpu()[..., 1], want[..., 1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,S,H,hd", [(2, 69, 4, 64), (2, 201, 4, 128), (3, 21, 2, 64), (1, 165, 4, 128), (2, 32, 1, 128)])
def test_attention(hb, prec, B, S, H, hd):
    g = torch.Generator().manual_seed(3)
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, generator=g)
    q, k, v = qkv.double().split(d, dim=-1)
    q = q.view(B, S, H, hd).transpose(1, 2)
    k = k.view(B, S, H, hd).transpose(1, 2)
    v = v.view(B, S, H, hd).transpose(1, 2)
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B, S, d)
    got = hb.test_attention(prec, qkv.cuda(), H)
    assert torch.isfinite(got).all()
    assert _rel(got, ref) < {"f32": 2e-5, "f16x3": 3e-5, "bf16x3": 2e-4, "bf16": 3e-2}[prec]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,S,H,hd", [(2, 224, 2, 128), (2, 224, 2, 64), (1, 225, 2, 128), (2, 128, 2, 128), (2, 129, 2, 64), (3, 1, 2, 64),
                                      (2, 8, 1, 128), (2, 16, 2, 64), (2, 17, 2, 128), (1, 113, 4, 64), (2, 97, 1, 128), (1, 208, 4, 128),
                                      (5, 193, 3, 64)])
def test_attention_size_boundaries(hb, prec, B, S, H, hd):
    """The edges of the resident-K kernel's cases: 224 keys (its largest, 7 key blocks), 225 (first streaming size), 128 / 129 keys
    (4 | 7 key-block instantiations), a single key, sizes one past a 16-row tile and sizes whose padded length is or is not a
    multiple of 32 (the Vt block burst and the output slots differ), an odd head count (the XCD-affine pair order's remainder)."""
    g = torch.Generator().manual_seed(7)
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, generator=g)
    q, k, v = qkv.double().split(d, dim=-1)
    q = q.view(B, S, H, hd).transpose(1, 2)
    k = k.view(B, S, H, hd).transpose(1, 2)
    v = v.view(B, S, H, hd).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ v).transpose(1, 2).reshape(B, S, d)
    got = hb.test_attention(prec, qkv.cuda(), H)
    assert torch.isfinite(got).all()
    assert _rel(got, ref) < {"f32": 2e-5, "f16x3": 3e-5, "bf16x3": 2e-4, "bf16": 3e-2}[prec]


STREAMING = 0x200FFFFF  # tamf_set_gemm_tuning: selection bit 512 = the streaming (online-softmax) attention kernel


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,S,H,hd,tuning", [(1, 300, 2, 128, -1), (2, 260, 2, 64, -1),   # > 224 keys: only the streaming kernel serves them
                                             (2, 201, 4, 128, STREAMING), (3, 21, 2, 64, STREAMING)])  # ... and it stays the A/B partner
def test_attention_streaming_kernel(hb, prec, B, S, H, hd, tuning):
    """The resident-K kernel covers S <= 224 (every shape of the launchers); longer sequences and selection bit 512 run the
    round-2 streaming kernel, which must stay correct."""
    g = torch.Generator().manual_seed(5)
    d = H * hd
    qkv = torch.randn(B, S, 3 * d, generator=g)
    q, k, v = qkv.double().split(d, dim=-1)
    q = q.view(B, S, H, hd).transpose(1, 2)
    k = k.view(B, S, H, hd).transpose(1, 2)
    v = v.view(B, S, H, hd).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1) @ v).transpose(1, 2).reshape(B, S, d)
    hb.hooks().tamf_set_gemm_tuning(tuning)
    try:
        got = hb.test_attention(prec, qkv.cuda(), H)
    finally:
        hb.hooks().tamf_set_gemm_tuning(-1)
    assert torch.isfinite(got).all()
    assert _rel(got, ref) < {"f32": 2e-5, "f16x3": 3e-5, "bf16x3": 2e-4, "bf16": 3e-2}[prec]


@pytest.mark.parametrize("tuning", [-1, STREAMING])
def test_attention_online_softmax_rescale(hb, tuning):
    """A late key block carrying the row maximum: the streaming kernel's running-max rescale path (selection bit 512), and the
    exact two-pass softmax of the resident-K kernel on the same input."""
    B, S, H, hd = 1, 201, 1, 128
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, S, 3 * hd, generator=g)
    qkv[0, 190, hd : 2 * hd] = qkv[0, 7, 0:hd] * 4.0  # key 190 aligned with query 7 -> huge score in the last block
    q, k, v = qkv.double().split(hd, dim=-1)
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    ref = att @ v
    hb.hooks().tamf_set_gemm_tuning(tuning)
    try:
        got = hb.test_attention("f32", qkv.cuda(), H)
    finally:
        hb.hooks().tamf_set_gemm_tuning(-1)
    assert _rel(got, ref) < 2e-5


def test_philox_matches_oracle(hb):
    from oracle import mdm_oracle as O

    got = hb.test_philox(1234567890123, 40, 3, 4, 99, 64).cpu().numpy()
    ref = O.philox_normal(1234567890123, np.arange(40, 44), 3, 99, 64)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    assert abs(got.mean()) < 0.03 and abs(got.std() - 1) < 0.03


@pytest.mark.parametrize("prec,nominal", [("f32", 157.3), ("bf16", 2500.0), ("f16x3", 2500.0)])
def test_mfma_sustained_rate(hb, prec, nominal):
    """tamf_bench_mfma_rate: register-only MFMA loops on every SIMD; the rate is positive, below the nominal peak of the
    instruction (MI355X_MICROARCH.md) and, for the 16-bit shapes with random operands, well above half of it."""
    tf, mhz = hb.mfma_sustained_rate(prec, 300)
    print(f"sustained {prec}: {tf:.0f} TFLOP/s, implied sclk >= {mhz:.0f} MHz")
    assert 0.5 * nominal < tf < 1.02 * nominal
    assert 800 < mhz < 2500
