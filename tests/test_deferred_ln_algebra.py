"""CPU: the algebra behind the deferred LayerNorm of the 16-bit modes (csrc/tamf_device.h "Deferred LayerNorm"),
restated in numpy float64 - what tamf_finalize_weights folds into the weights and what the kernels' epilogues then compute must equal
the reference's post-LN layer (interaction_segment_mdm.py:63-70: x = LayerNorm(u), eps 1e-5, then a Linear over x).

  LN(u) . W^T + b  =  rstd[m] (u . W''^T)[m][n] + c2[n]        W'' = W diag(gamma) (I - 1 1^T / d),  c2 = W beta + b
  statistics of a row from its 32-column block partials (S_b, Q_b):  mean = sum S_b / d,  M2 = sum (Q_b + 32 (S_b / 32 - mean)^2)
  residual add:  u_next = ((u - mean) rstd gamma + (beta + bias)) + acc

The GPU parity tests check the kernels against the oracle; this file pins the identities themselves, on inputs with a large row mean."""
import numpy as np


def _layer_norm(u, gamma, beta, eps=1e-5):
    mean = u.mean(-1, keepdims=True)
    var = ((u - mean) ** 2).mean(-1, keepdims=True)  # biased, as torch.nn.LayerNorm
    return (u - mean) / np.sqrt(var + eps) * gamma + beta


def _fold(W, b, gamma, beta):
    """tamf_finalize_weights (csrc/tamf_hip.hip, `fold`): every row of W diag(gamma) minus its own mean; c2 = W beta + b"""
    Wg = W * gamma[None, :]
    return Wg - Wg.mean(axis=1, keepdims=True), W @ beta + b


def _rows(rng, M, d, dc):
    return rng.standard_normal((M, d)) * rng.uniform(0.2, 5.0, (M, 1)) + dc * rng.standard_normal((M, 1))


def test_centring_and_gain_fold_into_the_weight():
    rng = np.random.default_rng(0)
    M, d, N = 37, 512, 96
    for dc in (0.0, 3.0, 40.0):  # rows whose mean is up to ~10 x their spread
        u = _rows(rng, M, d, dc)
        gamma, beta = rng.uniform(0.2, 5.0, d), 2.0 + rng.uniform(-0.5, 0.5, d)
        W, b = rng.standard_normal((N, d)) * 0.04, rng.standard_normal(N) * 0.1
        ref = _layer_norm(u, gamma, beta) @ W.T + b
        W2, c2 = _fold(W, b, gamma, beta)
        assert np.abs(W2.sum(axis=1)).max() < 1e-12  # rows of zero sum: the row mean drops out of the product by itself
        mean = u.mean(-1, keepdims=True)
        rstd = 1.0 / np.sqrt(((u - mean) ** 2).mean(-1, keepdims=True) + 1e-5)
        got = rstd * (u @ W2.T) + c2  # the consumers' epilogue: one fma with a per-row factor
        assert np.abs(got - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), dc


def test_row_statistics_from_block_partials():
    """what the residual GEMM's epilogue leaves per 32-column block and how ln_stage combines it (tamf_device.h)"""
    rng = np.random.default_rng(1)
    for d in (128, 256, 512):
        u = _rows(rng, 23, d, 5.0)
        blocks = u.reshape(u.shape[0], d // 32, 32)
        S = blocks.sum(-1)
        Q = ((blocks - S[..., None] / 32.0) ** 2).sum(-1)  # sum of squares about the block's own mean
        mean = S.sum(-1) / d
        M2 = (Q + 32.0 * (S / 32.0 - mean[:, None]) ** 2).sum(-1)
        np.testing.assert_allclose(mean, u.mean(-1), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(M2 / d, u.var(-1), rtol=1e-11, atol=1e-12)


def test_residual_add_with_the_normalisation_of_its_input_deferred():
    """two sublayers of a post-LN layer in the deferred form (residual stream = un-normalised sums) against the reference's order"""
    rng = np.random.default_rng(2)
    M, d, ff = 19, 256, 512
    x0 = _rows(rng, M, d, 2.0)  # the encoder input: no LayerNorm in front of layer 0's attention block
    g1, b1, g2, b2 = (rng.uniform(0.2, 5.0, d), 2.0 + rng.uniform(-0.5, 0.5, d), rng.uniform(0.2, 5.0, d), rng.uniform(-1.0, 1.0, d))
    Wo, bo = rng.standard_normal((d, d)) * 0.04, rng.standard_normal(d) * 0.1
    W1, c1 = rng.standard_normal((ff, d)) * 0.04, rng.standard_normal(ff) * 0.1
    W2, c2 = rng.standard_normal((d, ff)) * 0.04, rng.standard_normal(d) * 0.1
    a = rng.standard_normal((M, d))  # stands for the attention output
    relu = lambda z: np.maximum(z, 0.0)  # (any pointwise activation: the identity does not depend on it)
    # reference order
    x1 = _layer_norm(x0 + (a @ Wo.T + bo), g1, b1)
    x2 = _layer_norm(x1 + (relu(x1 @ W1.T + c1) @ W2.T + c2), g2, b2)
    # deferred: u1 = x0 + out-proj (identity "LayerNorm" in front); FFN1 consumes u1 through the folded weight; FFN2's residual add
    # normalises u1 on the way; the consumer of u2 (here: the comparison) applies LayerNorm 2
    u1 = x0 + (a @ Wo.T + bo)
    m1 = u1.mean(-1, keepdims=True)
    r1 = 1.0 / np.sqrt(((u1 - m1) ** 2).mean(-1, keepdims=True) + 1e-5)
    W1f, c1f = _fold(W1, c1, g1, b1)
    h = relu(r1 * (u1 @ W1f.T) + c1f)
    u2 = ((u1 - m1) * r1 * g1 + (b1 + c2)) + h @ W2.T
    np.testing.assert_allclose(_layer_norm(u2, g2, b2), x2, rtol=1e-9, atol=1e-9)
