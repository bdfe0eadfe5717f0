"""GPU: parity at the BENCHMARKED shapes (BASELINE.json configs[1], [3] shard, [4]) - the tile-scheduling branches the
bench runs (persistent one-round grids, remainder slices, XCD tile arrangements, fused FFN row tiles) are only reached
at M = B * Sp = 13 312 rows, so they get their own oracle comparison here.

  (i)   arch_mdm_l, B = 64, T = 196: one denoiser evaluation on all 64 clips against oracle.denoiser_forward
  (ii)  the same shape, a complete 5-step supplied-noise DDPM loop against oracle.sample_loop
  (iii) the GEMM hooks at (13312, 2048, 512), (13312, 1536, 512), (13312, 512, 2048), (13312, 512, 512) against float64
  (iv)  clip i sampled inside the B = 64 batch is bit-identical to clip i sampled alone
  (v)   B = 32 and B = 48 (other tile remainders; config 3's per-GPU shard)
  (vi)  arch_refine, B = 64, T = 196 against oracle.refine_forward (config 4)

Tolerances: about 3x the error observed on MI355X (max abs, outputs are O(1)); see DESIGN.md section 2.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_hip_forward import FWD_TOL, LOOP_TOL, _make_ctx, _set_cond  # noqa: E402

PRECS = list(FWD_TOL)
B_FULL, T_FULL = 64, 196


def _sub(cond, sl):
    return {k: (v[sl] if isinstance(v, torch.Tensor) else list(v[sl])) for k, v in cond.items()}


def _oracle_loop5(fx, tag, T):
    """(draws, oracle.sample_loop result) of the 5-step supplied-noise loop at B = 64: computed on first use, kept in the module fixture"""
    if "loop5" not in fx:
        from oracle import det
        from oracle import mdm_oracle as O

        shape = (B_FULL, 99, 1, T)
        draws = torch.from_numpy(np.stack([det.det_normal(det.step_noise_tag(tag, k), shape) for k in range(6)]))
        with torch.no_grad():
            ref = O.sample_loop(fx["sd"], fx["arch"], O.make_tables(5, "cosine"), fx["cond"], shape, lambda k: draws[k])
        fx["loop5"] = (draws, ref)
    return fx["loop5"]


@pytest.fixture(scope="module")
def full():
    """weights, conditioning, input and the oracle's outputs for the bench shape (computed once: ~5 s per forward)"""
    from oracle import mdm_oracle as O

    arch = O.ARCH_MDM_L
    sd = O.det_state_dict(arch, tag="full/w")
    cond = O.det_cond(B_FULL, T_FULL, tag="full/c", arch=arch)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B_FULL, 99, 1, T_FULL, generator=g)
    t = torch.full((B_FULL,), 500, dtype=torch.long)
    with torch.no_grad():
        ref = O.denoiser_forward(sd, arch, x, t, cond)
    return dict(arch=arch, sd=sd, cond=cond, x=x, t=t, ref=ref)


@pytest.mark.parametrize("prec", PRECS)
def test_forward_b64_t196_vs_oracle(full, prec):
    ctx = _make_ctx(full["arch"], full["sd"], B_FULL, T_FULL, prec)
    _set_cond(ctx, full["cond"])
    out = ctx.denoise(full["x"], full["t"]).cpu()
    assert torch.isfinite(out).all()
    per_clip = (out - full["ref"]).abs().amax(dim=(1, 2, 3))
    err = float(per_clip.max())
    print(f"fullsize forward[{prec}] B=64 T=196: max|err| = {err:.3e} (worst clip {int(per_clip.argmax())})")
    assert err < FWD_TOL[prec], (prec, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B", [32, 48])
def test_forward_other_batch_sizes(full, prec, B):
    """B = 32 is config 3's per-GPU shard, B = 48 a batch that is not a multiple of the row-tile grouping; clips are
    independent, so the reference is the first B clips of the B = 64 evaluation."""
    ctx = _make_ctx(full["arch"], full["sd"], B, T_FULL, prec)
    _set_cond(ctx, _sub(full["cond"], slice(0, B)))
    out = ctx.denoise(full["x"][:B], full["t"][:B]).cpu()
    err = float((out - full["ref"][:B]).abs().max())
    print(f"fullsize forward[{prec}] B={B}: max|err| = {err:.3e}")
    assert err < FWD_TOL[prec], (prec, B, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_loop5_b64_t196_vs_oracle(full, prec):
    from oracle import det
    from oracle import mdm_oracle as O

    N = 5
    draws, ref = _oracle_loop5(full, "full/eps", T_FULL)  # (the oracle's 5 steps at B = 64 take 20 s on the host: once per module, not per mode)
    ctx = _make_ctx(full["arch"], full["sd"], B_FULL, T_FULL, prec, n_steps=N)
    _set_cond(ctx, full["cond"])
    out = ctx.sample_loop(noise=draws).cpu()
    err = float((out - ref).abs().max())
    print(f"fullsize 5-step loop[{prec}]: max|err| = {err:.3e}")
    assert err < LOOP_TOL[prec], (prec, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_clip_in_b64_equals_clip_alone(full, prec):
    """Sharding invariance at the bench shape: a clip's Philox-noise sample does not depend on the batch around it."""
    N = 3
    ctx = _make_ctx(full["arch"], full["sd"], B_FULL, T_FULL, prec, n_steps=N)
    _set_cond(ctx, full["cond"])
    whole = ctx.sample_loop(noise=None, seed=7, clip_id_base=1000).cpu()
    for i in (0, 37, 63):
        _set_cond(ctx, _sub(full["cond"], slice(i, i + 1)))
        one = ctx.sample_loop(noise=None, seed=7, clip_id_base=1000 + i).cpu()
        assert torch.equal(one[0], whole[i]), (prec, i, float((one[0] - whole[i]).abs().max()))
    _set_cond(ctx, _sub(full["cond"], slice(32, 64)))
    half = ctx.sample_loop(noise=None, seed=7, clip_id_base=1032).cpu()
    assert torch.equal(half, whole[32:])
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B", [1, 7, 65, 100, 128])
def test_batches_beyond_and_beside_the_bench_shape(full, prec, B):
    """More clips per GPU than the bench shape (65, 100, 128: more than one round of the persistent grids, tile counts that do not
    divide by the CU count) and very small batches at T = 196: a context built for B clips evaluates them against the oracle
    (clips repeat the B = 64 fixture cyclically) and every clip's bits equal those of the same clip in the B = 64 batch."""
    idx = [i % B_FULL for i in range(B)]
    cond = {k: (v[idx] if isinstance(v, torch.Tensor) else [v[i] for i in idx]) for k, v in full["cond"].items()}
    x, t = full["x"][idx], full["t"][idx]
    ctx = _make_ctx(full["arch"], full["sd"], max(B, B_FULL), T_FULL, prec)
    _set_cond(ctx, cond)
    out = ctx.denoise(x, t).cpu()
    err = float((out - full["ref"][idx]).abs().max())
    print(f"forward[{prec}] B={B} T=196: max|err| = {err:.3e}")
    assert err < FWD_TOL[prec], (prec, B, err)
    _set_cond(ctx, full["cond"])
    base = ctx.denoise(full["x"], full["t"]).cpu()
    assert torch.equal(out, base[idx]), (prec, B, float((out - base[idx]).abs().max()))
    ctx.close()


GEMM_TOL = {"f32": 1e-5, "f16x3": 2e-5, "bf16x3": 1e-4, "bf16": 2e-2}


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("N,K", [(2048, 512), (1536, 512), (512, 2048), (512, 512)])
def test_gemm_bench_shapes(prec, N, K):
    """The launcher branches of M = 13 312: persistent one-round grid (1 248 QKV tiles), full rounds + slices (FFN1),
    the XCD tile arrangements."""
    from oakink2_tamf_amd import hip_backend as hb

    M = 13312
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    got = hb.test_gemm(prec, a.cuda(), w.cuda(), b.cuda(), 0).double().cpu()
    ref = a.double() @ w.double().t() + b.double()
    rel = float((got - ref).abs().max() / ref.abs().max())
    assert rel < GEMM_TOL[prec], (prec, N, K, rel)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("K", [512, 2048])
def test_gemm_resid_bench_shapes(prec, K):
    """the residual GEMMs (out-proj K = 512, FFN2 K = 2048) at the bench shape, on the whole-clip tiles, with the LayerNorm of the residual deferred"""
    from oakink2_tamf_amd import hip_backend as hb

    M, N = 13312, 512
    g = torch.Generator().manual_seed(2)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bb = torch.randn(N, generator=g) * 0.1
    u = torch.randn(M, N, generator=g) * 2.0 + 0.7
    ga = 1 + 0.1 * torch.randn(N, generator=g)
    ref = (torch.nn.functional.layer_norm(u.double(), (N,), ga.double(), None, 1e-5) + bb.double()) + a.double() @ w.double().t()
    got, st = hb.test_gemm_resid(prec, a.cuda(), w.cuda(), bb.cuda(), ga.cuda(), u.cuda(), hb.block_stats(u).cuda())
    got = got.double().cpu()
    rel = float((got - ref).abs().max() / ref.abs().max())
    assert rel < GEMM_TOL[prec], (prec, K, rel)
    want = hb.block_stats(got.float())
    assert torch.allclose(st.cpu(), want, rtol=1e-4, atol=1e-3)


REFINE_TOL = {"f32": 3e-5, "f16x3": 3e-5, "bf16x3": 2e-4, "bf16": 1e-1}


@pytest.mark.parametrize("prec", PRECS)
def test_refine_b64_t196_vs_oracle(prec):
    """BASELINE.json configs[3]: the R trunk at B = 64, T = 196 (synthetic h2o_dist) against oracle.refine_forward."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_REFINE
    sd = O.det_state_dict(arch, tag="fullr/w")
    cond = O.det_cond(B_FULL, T_FULL, tag="fullr/c", arch=arch)
    g = torch.Generator().manual_seed(3)
    x_in = torch.randn(B_FULL, T_FULL, 99, generator=g)
    h2o = torch.rand(B_FULL, T_FULL, 778, generator=g) * 0.2
    with torch.no_grad():
        ref = O.refine_forward(sd, arch, x_in, h2o, cond)
    ctx = _make_ctx(arch, sd, B_FULL, T_FULL, prec)
    _set_cond(ctx, cond)
    out = ctx.refine(x_in, h2o).cpu()
    err = float((out - ref).abs().max())
    print(f"fullsize refine[{prec}]: max|err| = {err:.3e}")
    assert err < REFINE_TOL[prec], (prec, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("N", [1536, 2048, 512])
def test_gemm_clip_tiles_exact_integers(prec, N):
    """Small-integer operands are exact in every arithmetic mode, so the bench-shape launches - the clip-aligned tiles (one
    M tile = the 208 rows of one clip; 256- / 128-column tiles for N = 2048 / 512) and the persistent 128 x 128 grid
    (N = 1536) - must reproduce the integer product bit for bit: any slip in the tile -> (clip, column) map, the slab-wise
    epilogue or the XCD remaps shows up as a wrong integer."""
    from oakink2_tamf_amd import hip_backend as hb

    M, K = 13312, 128
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    b = torch.randint(-8, 9, (N,), generator=g).float()
    got = hb.test_gemm(prec, a.cuda(), w.cuda(), b.cuda(), 0).cpu()
    ref = (a.long() @ w.long().t() + b.long()).float()
    assert torch.equal(got, ref), (prec, N, int((got != ref).sum()))


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_forward_clip_tiles_with_padded_rows(prec):
    """T = 190 -> S = 195, padded to 200 rows per clip: the 208-row clip tile clamps its last 8 staged rows; B = 48 clips
    (1.5 rounds of the 256 CUs).  Against the oracle on all clips."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_MDM_L
    B, T = 48, 190
    sd = O.det_state_dict(arch, tag="full/w")
    cond = O.det_cond(B, T, tag="pad/c", arch=arch)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 99, 1, T, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        ref = O.denoiser_forward(sd, arch, x, t, cond)
    ctx = _make_ctx(arch, sd, B, T, prec)
    _set_cond(ctx, cond)
    out = ctx.denoise(x, t).cpu()
    err = float((out - ref).abs().max())
    print(f"padded-rows forward[{prec}] B=48 T=190: max|err| = {err:.3e}")
    assert err < FWD_TOL[prec], (prec, err)
    ctx.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16"])
@pytest.mark.parametrize("B,T", [(64, 174), (64, 147), (32, 188)])
def test_forward_clip_tiles_at_their_largest_padding(prec, B, T):
    """The clip tiles stage the rows of a clip with the per-lane row clamp only on their last four 8-row pieces (csrc/tamf_gemm_clip.h,
    clip_issue): the shapes with the MOST padding rows each tile family accepts - T = 174: Sp = 184 of a 208-row tile (24 padding rows);
    T = 147: Sp = 152 of a 176-row tile (24); B = 32, T = 188: Sp = 200 as row parts 112 + 88 (24 in the second part).  Against the
    oracle on all clips."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_MDM_L
    sd = O.det_state_dict(arch, tag="full/w")
    cond = O.det_cond(B, T, tag="maxpad/c", arch=arch)
    g = torch.Generator().manual_seed(1000 + T)
    x = torch.randn(B, 99, 1, T, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    with torch.no_grad():
        ref = O.denoiser_forward(sd, arch, x, t, cond)
    ctx = _make_ctx(arch, sd, B, T, prec)
    _set_cond(ctx, cond)
    out = ctx.denoise(x, t).cpu()
    err = float((out - ref).abs().max())
    print(f"max-padding forward[{prec}] B={B} T={T}: max|err| = {err:.3e}")
    assert torch.isfinite(out).all() and err < FWD_TOL[prec], (prec, B, T, err)
    ctx.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16"])
@pytest.mark.parametrize("B,arch_name", [(64, "ARCH_MDM_L"), (32, "ARCH_MDM_L"), (64, "ARCH_MDM")])
def test_clip_tiles_over_the_clip_lengths_they_accept(prec, B, arch_name, test_hooks):
    """27 clip lengths between T = 139 and 204 (every fourth one plus the edges of the tile families; Sp = 144 .. 216: below, inside and
    above the 176- / 208-row clip tiles and their row parts): the default kernels against the 128 x 128 tiles (selection 1: no clip
    tiles at all) - the same bits - and finite.  (T = 172 .. 187 and 140 .. 155 used to run the f32 V^T clip tile past the V^T rows:
    vt_row_keys, csrc/tamf_hip.hip.  A comparison of kernels with kernels cannot see a store that lands in a neighbouring live buffer;
    EVERY length 1 .. 224 is walked by tests/test_hip_guardbands.py with guard bands around every allocation.)"""
    from oracle import mdm_oracle as O
    from oakink2_tamf_amd.hip_backend import lib

    arch = getattr(O, arch_name)  # (ARCH_MDM: the reference's smaller denoiser - other GEMM widths, 64-wide heads)
    sd = O.det_state_dict(arch, tag="full/w")
    bad = []
    lengths = list(range(139, 205, 4)) + [147, 155, 156, 171, 172, 179, 187, 188, 203, 204]
    # ONE context for all lengths (the kernels are chosen by the shape of the call, not by the context's capacity; a context per length
    # spent most of this test uploading and repacking 109 MB of weights 27 times)
    ctx = _make_ctx(arch, sd, B, max(lengths), prec)
    try:
        for T in lengths:
            cond = O.det_cond(B, T, tag="sweep/c", arch=arch)
            g = torch.Generator().manual_seed(T)
            x = torch.randn(B, 99, 1, T, generator=g)
            t = torch.randint(0, 1000, (B,), generator=g)
            _set_cond(ctx, cond)
            lib().tamf_set_gemm_tuning(-1)
            a = ctx.denoise(x, t).cpu()
            lib().tamf_set_gemm_tuning((1 << 20) | 0xFFFFF)
            b = ctx.denoise(x, t).cpu()
            lib().tamf_set_gemm_tuning(-1)
            if not (torch.isfinite(a).all() and torch.equal(a, b)):
                bad.append((T, float((a - b).abs().max())))
    finally:
        lib().tamf_set_gemm_tuning(-1)
        ctx.close()
    assert not bad, (prec, B, bad)


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16"])
@pytest.mark.parametrize("T", [196, 160])
def test_kernel_choice_over_batch_sizes(prec, T, test_hooks):
    """The launch rules switch kernels with the batch size (clip tiles from 74 % fill, row-part tiles up to half the CUs, query splits of the
    attention, 64-row tiles for short-K GEMMs, row-part tiles of the residual GEMMs at 32 clips or fewer): at 24 batch sizes from 1 to 128 the default
    kernels against selection 1 (no clip tiles anywhere) - the same bits - and clip 0 of every batch against clip 0 alone."""
    from oracle import mdm_oracle as O
    from oakink2_tamf_amd.hip_backend import lib

    arch = O.ARCH_MDM_L
    sd = O.det_state_dict(arch, tag="full/w")
    BMAX = 128
    cond = O.det_cond(BMAX, T, tag="bsweep/c", arch=arch)
    g = torch.Generator().manual_seed(7 + T)
    x = torch.randn(BMAX, 99, 1, T, generator=g)
    t = torch.randint(0, 1000, (BMAX,), generator=g)
    bad = []
    alone = None
    ctx = _make_ctx(arch, sd, BMAX, T, prec)  # (one context: see test_clip_tiles_over_the_clip_lengths_they_accept)
    try:
        for B in [1, 2, 3, 5, 8, 13, 16, 21, 24, 31, 32, 33, 40, 47, 48, 49, 56, 63, 64, 65, 72, 96, 127, 128]:
            _set_cond(ctx, _sub(cond, slice(0, B)))
            lib().tamf_set_gemm_tuning(-1)
            a = ctx.denoise(x[:B], t[:B]).cpu()
            lib().tamf_set_gemm_tuning((1 << 20) | 0xFFFFF)
            b = ctx.denoise(x[:B], t[:B]).cpu()
            lib().tamf_set_gemm_tuning(-1)
            if alone is None:
                alone = a[0].clone()
            if not (torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a[0], alone)):
                bad.append((B, float((a - b).abs().max()), float((a[0] - alone).abs().max())))
    finally:
        lib().tamf_set_gemm_tuning(-1)
        ctx.close()
    assert not bad, (prec, T, bad)


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16"])
def test_refine_and_loop_over_clip_lengths(prec, test_hooks):
    """The R trunk (3 prefix tokens: S = T + 3) and the hipGraph sampling loop of G (20 steps, device Philox) over clip lengths around
    the clip tiles' limits, B = 64: default kernels against selection 1 (no clip tiles) - the same bits, finite."""
    from oracle import mdm_oracle as O
    from oakink2_tamf_amd.hip_backend import lib

    bad = []
    try:
        arch = O.ARCH_REFINE
        sd = O.det_state_dict(arch, tag="fullr/w")
        for T in [141, 150, 158, 160, 173, 174, 176, 181, 189, 190, 196, 205]:
            cond = O.det_cond(64, T, tag="rsweep/c", arch=arch)
            g = torch.Generator(device="cuda").manual_seed(T)  # (10 M normals per length: drawn on the device)
            x_in = torch.randn(64, T, 99, generator=g, device="cuda")
            h2o = torch.rand(64, T, 778, generator=g, device="cuda") * 0.2
            ctx = _make_ctx(arch, sd, 64, T, prec)
            _set_cond(ctx, cond)
            lib().tamf_set_gemm_tuning(-1)
            a = ctx.refine(x_in, h2o).cpu()
            lib().tamf_set_gemm_tuning((1 << 20) | 0xFFFFF)
            b = ctx.refine(x_in, h2o).cpu()
            lib().tamf_set_gemm_tuning(-1)
            ctx.close()
            if not (torch.isfinite(a).all() and torch.equal(a, b)):
                bad.append(("refine", T, float((a - b).abs().max())))
        arch = O.ARCH_MDM_L
        sd = O.det_state_dict(arch, tag="full/w")
        tab = O.make_tables(20, "cosine")
        for T in [147, 160, 174, 187, 188, 196]:
            cond = O.det_cond(64, T, tag="lsweep/c", arch=arch)
            outs = []
            for tune in (-1, (1 << 20) | 0xFFFFF):
                lib().tamf_set_gemm_tuning(tune)
                ctx = _make_ctx(arch, sd, 64, T, prec)
                _set_cond(ctx, cond)
                ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
                out = torch.empty(64, 99, 1, T, device="cuda")
                ctx.sample_loop(seed=5, out=out)
                outs.append(out.cpu())
                ctx.close()
            lib().tamf_set_gemm_tuning(-1)
            if not (torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])):
                bad.append(("loop", T, float((outs[0] - outs[1]).abs().max())))
    finally:
        lib().tamf_set_gemm_tuning(-1)
    assert not bad, (prec, bad)


def test_modes_agree_on_random_shapes():
    """48 random (B, T) shapes, B in [1, 130], T in [8, 204] - most of them shapes no other test visits: the four arithmetic modes run
    different kernels for the same launch (f32: clip tiles for QKV and out-proj, its own LayerNorm forms; bf16: LayerNorm fused into
    FFN2; the split modes: neither), so a slip in one mode's path shows as a disagreement far above the modes' rounding differences."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_MDM_L
    sd = O.det_state_dict(arch, tag="full/w")
    rng = np.random.RandomState(20261003)
    shapes = [(int(rng.randint(1, 131)), int(rng.randint(8, 205))) for _ in range(48)]
    tol = {"f16x3": 3e-5, "bf16x3": 2e-4, "bf16": 6e-2}
    bad = []
    # one context per mode for all 48 shapes (192 contexts = 192 weight uploads before)
    ctxs = {prec: _make_ctx(arch, sd, max(b for b, _ in shapes), max(t_ for _, t_ in shapes), prec) for prec in ["f32", "f16x3", "bf16x3", "bf16"]}
    for B, T in shapes:
        cond = O.det_cond(B, T, tag="rand/c", arch=arch)
        g = torch.Generator().manual_seed(B * 1000 + T)
        x = torch.randn(B, 99, 1, T, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        outs = {}
        for prec, ctx in ctxs.items():
            _set_cond(ctx, cond)
            outs[prec] = ctx.denoise(x, t).cpu()
        for prec, tl in tol.items():
            d = float((outs[prec] - outs["f32"]).abs().max())
            if not (torch.isfinite(outs[prec]).all() and d < tl):
                bad.append((B, T, prec, d))
        if not torch.isfinite(outs["f32"]).all():
            bad.append((B, T, "f32", float("nan")))
    for ctx in ctxs.values():
        ctx.close()
    assert not bad, bad


# selection overrides of tamf_set_gemm_tuning (bits 20..): every alternative kernel of a launch must give the SAME BITS as the default
# one - that is what makes a clip's sample independent of the batch it is in (different batch sizes select different kernels)
SELECTIONS = {
    0x001: "no clip tiles at all (128 x 128 tiles everywhere)",
    0x002: "FFN2 / out-proj not on clip tiles",
    0x008: "FFN1 on the 128 x 128 tiles",
    0x010: "residual GEMMs of a few clips without the 32- / 64-row tiles",
    0x040: "QKV on the 128 x 128 tiles (f32 default: clip tiles, Q|K and V transposed)",
    0x400: "clip tiles from 50 % utilisation",
}


@pytest.mark.parametrize("prec", PRECS)
def test_kernel_selections_give_the_same_bits(full, prec, test_hooks):
    from oakink2_tamf_amd.hip_backend import lib

    B = 48  # (1.5 rounds of clip tiles for FFN2: remainders too)
    ctx = _make_ctx(full["arch"], full["sd"], B, T_FULL, prec)
    _set_cond(ctx, _sub(full["cond"], slice(0, B)))
    try:
        ref = ctx.denoise(full["x"][:B], full["t"][:B]).cpu()
        for sel, what in SELECTIONS.items():
            lib().tamf_set_gemm_tuning((sel << 20) | 0xFFFFF)
            got = ctx.denoise(full["x"][:B], full["t"][:B]).cpu()
            assert torch.equal(got, ref), (prec, hex(sel), what, float((got - ref).abs().max()))
    finally:
        lib().tamf_set_gemm_tuning(-1)
        ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_ffn1_column_split_rounds_give_the_same_bits(full, prec, test_hooks):
    """Selection bit 32 (round 6): FFN1's two rounds of whole-clip tiles at B = 64 split by COLUMNS (every clip's column tiles 0 - 3, then
    4 - 7) instead of by clips - another order of the same tiles: the same bits, against the oracle too."""
    from oakink2_tamf_amd.hip_backend import lib

    ctx = _make_ctx(full["arch"], full["sd"], B_FULL, T_FULL, prec)
    _set_cond(ctx, full["cond"])
    try:
        ref = ctx.denoise(full["x"], full["t"]).cpu()
        lib().tamf_set_gemm_tuning((0x020 << 20) | 0xFFFFF)
        got = ctx.denoise(full["x"], full["t"]).cpu()
        assert torch.equal(got, ref), (prec, float((got - ref).abs().max()))
        assert float((got - full["ref"]).abs().max()) < FWD_TOL[prec]
    finally:
        lib().tamf_set_gemm_tuning(-1)
        ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_small_and_mid_batch_tiles_give_the_same_bits(full, full160, prec, test_hooks):
    """Calls of a few clips run their GEMMs on gemm_deep_kernel (csrc/tamf_gemm_deep.h: 32 x 64 / 32 x 128 / 64 x 128 tiles with a 3- to
    6-stage K pipeline; one clip per call is the reference launcher's own pattern, launch/sample.py:202-229), 20 - 39 clips of the 16-bit
    modes their residual GEMMs: same bits as the tiles of the big batches (selection bit 16), and within the tolerance of the oracle."""
    from oakink2_tamf_amd.hip_backend import lib

    for fx, T, Bs in ((full, T_FULL, (1, 2, 5, 9, 17, 24, 33)), (full160, T_DS, (1, 3, 6, 12, 30))):
        for B in Bs:
            cond, x, t = _sub(fx["cond"], slice(0, B)), fx["x"][:B], fx["t"][:B]
            ctx = _make_ctx(fx["arch"], fx["sd"], B, T, prec)
            _set_cond(ctx, cond)
            try:
                ref = ctx.denoise(x, t).cpu()
                lib().tamf_set_gemm_tuning((0x010 << 20) | 0xFFFFF)
                got = ctx.denoise(x, t).cpu()
            finally:
                lib().tamf_set_gemm_tuning(-1)
                ctx.close()
            assert torch.equal(got, ref), (prec, B, T, float((got - ref).abs().max()))
            err = float((ref - fx["ref"][:B]).abs().max())
            assert err < FWD_TOL[prec], (prec, B, T, err)


# ---- T = 160: the only clip length the reference's dataset emits (dataset/interaction_segment.py:291, slice_max_len = 160) ------------
# S = 165, padded to Sp = 168 rows = 10.5 MFMA row tiles: clip tiles of 11 row tiles (6 + 5 as row parts at 32 clips per GPU), the
# resident-K attention with 12 key tiles, 64-row LayerNorm blocks that straddle clips, V^T rows of 192 keys.
T_DS = 160


@pytest.fixture(scope="module")
def full160():
    from oracle import mdm_oracle as O

    arch = O.ARCH_MDM_L
    sd = O.det_state_dict(arch, tag="full/w")
    cond = O.det_cond(B_FULL, T_DS, tag="full160/c", arch=arch)
    g = torch.Generator().manual_seed(160)
    x = torch.randn(B_FULL, 99, 1, T_DS, generator=g)
    t = torch.randint(0, 1000, (B_FULL,), generator=g)
    with torch.no_grad():
        ref = O.denoiser_forward(sd, arch, x, t, cond)
    return dict(arch=arch, sd=sd, cond=cond, x=x, t=t, ref=ref)


@pytest.mark.parametrize("prec", PRECS)
def test_forward_b64_t160_vs_oracle(full160, prec):
    ctx = _make_ctx(full160["arch"], full160["sd"], B_FULL, T_DS, prec)
    _set_cond(ctx, full160["cond"])
    out = ctx.denoise(full160["x"], full160["t"]).cpu()
    assert torch.isfinite(out).all()
    per_clip = (out - full160["ref"]).abs().amax(dim=(1, 2, 3))
    err = float(per_clip.max())
    print(f"fullsize forward[{prec}] B=64 T=160: max|err| = {err:.3e} (worst clip {int(per_clip.argmax())})")
    assert err < FWD_TOL[prec], (prec, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_loop5_b64_t160_vs_oracle(full160, prec):
    from oracle import det
    from oracle import mdm_oracle as O

    N = 5
    draws, ref = _oracle_loop5(full160, "full160/eps", T_DS)
    ctx = _make_ctx(full160["arch"], full160["sd"], B_FULL, T_DS, prec, n_steps=N)
    _set_cond(ctx, full160["cond"])
    out = ctx.sample_loop(noise=draws).cpu()
    err = float((out - ref).abs().max())
    print(f"fullsize 5-step loop[{prec}] T=160: max|err| = {err:.3e}")
    assert err < LOOP_TOL[prec], (prec, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_clip_in_b64_t160_equals_clip_alone(full160, prec):
    """Sharding invariance at T = 160: clip i of the B = 64 batch (11-row-tile clip tiles) vs alone (128 x 128 tiles) vs inside
    the B = 32 half (6 + 5 row-part tiles, attention on split query ranges) - bit for bit."""
    N = 3
    ctx = _make_ctx(full160["arch"], full160["sd"], B_FULL, T_DS, prec, n_steps=N)
    _set_cond(ctx, full160["cond"])
    whole = ctx.sample_loop(noise=None, seed=7, clip_id_base=1000).cpu()
    for i in (0, 21, 63):
        _set_cond(ctx, _sub(full160["cond"], slice(i, i + 1)))
        one = ctx.sample_loop(noise=None, seed=7, clip_id_base=1000 + i).cpu()
        assert torch.equal(one[0], whole[i]), (prec, i, float((one[0] - whole[i]).abs().max()))
    _set_cond(ctx, _sub(full160["cond"], slice(32, 64)))
    half = ctx.sample_loop(noise=None, seed=7, clip_id_base=1032).cpu()
    assert torch.equal(half, whole[32:])
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B", [32, 48])
def test_forward_other_batch_sizes_t160(full160, prec, B):
    ctx = _make_ctx(full160["arch"], full160["sd"], B, T_DS, prec)
    _set_cond(ctx, _sub(full160["cond"], slice(0, B)))
    out = ctx.denoise(full160["x"][:B], full160["t"][:B]).cpu()
    err = float((out - full160["ref"][:B]).abs().max())
    print(f"fullsize forward[{prec}] B={B} T=160: max|err| = {err:.3e}")
    assert err < FWD_TOL[prec], (prec, B, err)
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_kernel_selections_give_the_same_bits_t160(full160, prec, test_hooks):
    from oakink2_tamf_amd.hip_backend import lib

    B = 48
    ctx = _make_ctx(full160["arch"], full160["sd"], B, T_DS, prec)
    _set_cond(ctx, _sub(full160["cond"], slice(0, B)))
    try:
        ref = ctx.denoise(full160["x"][:B], full160["t"][:B]).cpu()
        for sel, what in SELECTIONS.items():
            lib().tamf_set_gemm_tuning((sel << 20) | 0xFFFFF)
            got = ctx.denoise(full160["x"][:B], full160["t"][:B]).cpu()
            assert torch.equal(got, ref), (prec, hex(sel), what, float((got - ref).abs().max()))
    finally:
        lib().tamf_set_gemm_tuning(-1)
        ctx.close()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("clips", [64, 32])
def test_gemm_clip_tiles_exact_integers_t160(prec, clips):
    """The 11-row-tile clip tiles (64 clips) and their 6 + 5 row parts (32 clips) on M = clips x 168 rows: exact integers, so any slip
    in the clamped half row tile, the part boundaries or the tile -> (clip, column) map is a wrong integer."""
    from oakink2_tamf_amd import hip_backend as hb

    M, N, K = clips * 168, 512, 128
    g = torch.Generator().manual_seed(6)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    b = torch.randint(-8, 9, (N,), generator=g).float()
    got = hb.test_gemm(prec, a.cuda(), w.cuda(), b.cuda(), 0).cpu()
    ref = (a.long() @ w.long().t() + b.long()).float()
    assert torch.equal(got, ref), (prec, clips, int((got != ref).sum()))


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_forward_arch_mdm_b64_t160_vs_oracle(prec):
    """The small architecture (arch_mdm: d = 256, ff = 1024, 4 heads of 64) at the dataset's shape, B = 64, T = 160: the 11-row-tile clip
    tiles at K = 256 / N = 1024, the 12-key-tile resident-K attention at hd = 64, the 256-column LayerNorm tiles."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_MDM
    sd = O.det_state_dict(arch, tag="mdm160/w")
    cond = O.det_cond(B_FULL, T_DS, tag="mdm160/c", arch=arch)
    g = torch.Generator().manual_seed(161)
    x = torch.randn(B_FULL, 99, 1, T_DS, generator=g)
    t = torch.randint(0, 1000, (B_FULL,), generator=g)
    with torch.no_grad():
        ref = O.denoiser_forward(sd, arch, x, t, cond)
    ctx = _make_ctx(arch, sd, B_FULL, T_DS, prec)
    _set_cond(ctx, cond)
    out = ctx.denoise(x, t).cpu()
    err = float((out - ref).abs().max())
    print(f"arch_mdm forward[{prec}] B=64 T=160: max|err| = {err:.3e}")
    assert err < FWD_TOL[prec], (prec, err)
    # a clip alone = the clip in the batch, bit for bit (other kernel selections at B = 1)
    _set_cond(ctx, _sub(cond, slice(5, 6)))
    one = ctx.denoise(x[5:6], t[5:6]).cpu()
    assert torch.equal(one[0], out[5])
    ctx.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_refine_b64_t160_vs_oracle(prec):
    """The R trunk (arch_refine) at the dataset's clip length, B = 64, T = 160 (3 prefix tokens: S = 163, Sp = 168)."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_REFINE
    sd = O.det_state_dict(arch, tag="fullr/w")
    cond = O.det_cond(B_FULL, T_DS, tag="fullr160/c", arch=arch)
    g = torch.Generator().manual_seed(31)
    x_in = torch.randn(B_FULL, T_DS, 99, generator=g)
    h2o = torch.rand(B_FULL, T_DS, 778, generator=g) * 0.2
    with torch.no_grad():
        ref = O.refine_forward(sd, arch, x_in, h2o, cond)
    ctx = _make_ctx(arch, sd, B_FULL, T_DS, prec)
    _set_cond(ctx, cond)
    out = ctx.refine(x_in, h2o).cpu()
    err = float((out - ref).abs().max())
    print(f"refine[{prec}] B=64 T=160: max|err| = {err:.3e}")
    assert err < REFINE_TOL[prec], (prec, err)
    ctx.close()
