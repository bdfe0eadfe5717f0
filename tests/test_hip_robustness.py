"""GPU: robustness of the C-ABI path - odd shapes, re-use of a context across batches, determinism, multiple contexts."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ctx(arch, sd, B, T, prec="f32", n_steps=8):
    from test_hip_forward import _make_ctx

    return _make_ctx(arch, sd, B, T, prec, n_steps=n_steps)


def _cond(B, T, tag, nobj=2):
    from oracle import mdm_oracle as O

    return O.det_cond(B, T, nobj=nobj, tag=tag, arch=O.ARCH_TINY)


@pytest.mark.parametrize("B,T,nobj", [(1, 16, 1), (1, 3, 2), (5, 27, 4), (2, 59, 1), (3, 123, 2), (1, 1, 1), (2, 219, 1), (2, 220, 2), (1, 400, 1)])
def test_odd_shapes_against_oracle(B, T, nobj):
    """Sp = round_up(T + 5, 8) and Skp = round_up(T + 5, 32) padding paths, single clip, single frame tile, 4 objects; a single frame;
    T = 219 / 220 (224 keys: the largest resident-K attention case, 225: the first one of the streaming kernel) and a long clip."""
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd = O.det_state_dict(arch, tag="rob/w")
    cond = _cond(B, T, f"rob/{B}/{T}", nobj)
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = torch.randn(B, 99, 1, T, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    ref = O.denoiser_forward(sd, arch, x, t, cond)
    for prec, tol in (("f32", 1e-5), ("f16x3", 1e-5), ("bf16x3", 6e-5)):
        ctx = _ctx(arch, sd, B, T, prec)
        _set_cond(ctx, cond)
        out = ctx.denoise(x, t).cpu()
        assert float((out - ref).abs().max()) < tol, (prec, B, T)
        ctx.close()


def test_context_reuse_smaller_batches_and_determinism():
    """One context (max B=4, T=32) serves smaller (B, T) batches; the same call twice is bit-identical; the graph is
    re-captured when the batch shape or the noise source changes."""
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd = O.det_state_dict(arch, tag="rob/w")
    ctx = _ctx(arch, sd, 4, 32, "bf16x3", n_steps=6)
    tab = O.make_tables(6, "cosine")
    outs = {}
    for (B, T) in ((4, 32), (2, 16), (3, 32), (4, 32)):
        cond = _cond(B, T, f"reuse/{B}/{T}")
        _set_cond(ctx, cond)
        a = ctx.sample_loop(noise=None, seed=5, clip_id_base=0).cpu()
        b = ctx.sample_loop(noise=None, seed=5, clip_id_base=0).cpu()
        assert torch.equal(a, b)
        ref = O.sample_loop(sd, arch, tab, cond, (B, 99, 1, T),
                            lambda k: torch.from_numpy(O.philox_normal(5, np.arange(B), k, 99, T)))
        assert float((a - ref).abs().max()) < 1e-3
        outs.setdefault((B, T), []).append(a)
        g = torch.Generator().manual_seed(1)
        draws = torch.randn(7, B, 99, 1, T, generator=g)
        c = ctx.sample_loop(noise=draws).cpu()
        refc = O.sample_loop(sd, arch, tab, cond, (B, 99, 1, T), lambda k: draws[k])
        assert float((c - refc).abs().max()) < 1e-3
    assert torch.equal(outs[(4, 32)][0], outs[(4, 32)][1])
    ctx.close()


def test_two_contexts_do_not_interfere():
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd1, sd2 = O.det_state_dict(arch, tag="two/a"), O.det_state_dict(arch, tag="two/b")
    c1, c2 = _ctx(arch, sd1, 2, 16), _ctx(arch, sd2, 2, 16)
    cond = _cond(2, 16, "two/c")
    _set_cond(c1, cond)
    _set_cond(c2, cond)
    x = torch.randn(2, 99, 1, 16, generator=torch.Generator().manual_seed(3))
    t = torch.tensor([4, 900])
    o1a = c1.denoise(x, t).cpu()
    o2 = c2.denoise(x, t).cpu()
    o1b = c1.denoise(x, t).cpu()
    assert torch.equal(o1a, o1b) and not torch.equal(o1a, o2)
    assert float((o1a - O.denoiser_forward(sd1, arch, x, t, cond)).abs().max()) < 2e-5
    assert float((o2 - O.denoiser_forward(sd2, arch, x, t, cond)).abs().max()) < 2e-5
    c1.close()
    c2.close()


def test_two_contexts_driven_from_two_threads():
    """Round 6: the product library takes no process-wide lock any more (it guarded the kernel-selection word of tamf_set_gemm_tuning,
    which now lives in the hooks build only).  Two contexts, two host threads, two streams, each thread running plain-launch and
    hipGraph loops and single evaluations of its own context at the same time: every result equals the one the same context gives alone."""
    import threading

    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sds = [O.det_state_dict(arch, tag="thr/a"), O.det_state_dict(arch, tag="thr/b")]
    precs = ["f16x3", "bf16"]
    ctxs = [_ctx(arch, sds[i], 3, 24, prec=precs[i], n_steps=40) for i in range(2)]
    conds = [_cond(3, 24, f"thr/c{i}") for i in range(2)]
    x = torch.randn(3, 99, 1, 24, generator=torch.Generator().manual_seed(3))
    t = torch.tensor([4, 900, 17])
    for c, cd in zip(ctxs, conds):
        _set_cond(c, cd)
    alone = [(c.sample_loop(seed=11 + i).cpu(), c.sample_loop(seed=11 + i, use_graph=False).cpu(), c.denoise(x, t).cpu()) for i, c in enumerate(ctxs)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    got, errs = [[], []], []

    def worker(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(6):
                    a = ctxs[i].sample_loop(seed=11 + i)
                    b = ctxs[i].sample_loop(seed=11 + i, use_graph=False)
                    c = ctxs[i].denoise(x, t)
                    streams[i].synchronize()
                    got[i].append((a.cpu(), b.cpu(), c.cpu()))
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for w in th:
        w.start()
    for w in th:
        w.join()
    assert not errs, errs
    for i in range(2):
        assert len(got[i]) == 6
        for a, b, c in got[i]:
            assert torch.equal(a, alone[i][0]) and torch.equal(b, alone[i][1]) and torch.equal(c, alone[i][2]), i
        assert torch.equal(alone[i][0], alone[i][1])  # (graph == plain launches)
    for c in ctxs:
        c.close()


def test_ddpm_step_entry_point():
    """tamf_ddpm_step reproduces the oracle's float32 arithmetic (same op order, no FMA contraction); the only source of
    difference is the last bit of sigma = exp(0.5 * logvar) (libm expf vs torch.exp), hence 1e-6 instead of bit equality;
    t = 0 (no noise term) is exact."""
    from oracle import mdm_oracle as O

    arch = O.ARCH_TINY
    ctx = _ctx(arch, O.det_state_dict(arch, tag="rob/w"), 2, 16, n_steps=1000)
    tab = O.make_tables(1000, "cosine")
    g = torch.Generator().manual_seed(9)
    xt, x0, nz = (torch.randn(2, 99, 1, 16, generator=g) for _ in range(3))
    for t in (0, 1, 500, 999):
        got = ctx.ddpm_step(xt, x0, t, nz).cpu()
        ref = O.ddpm_step(tab, xt, x0, t, nz)
        if t == 0:
            assert torch.equal(got, ref)
        assert float((got - ref).abs().max()) < 1e-6, t
    ctx.close()


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16x3", "bf16"])
def test_clip_sample_independent_of_batch_size_and_position(prec):
    """Sharding invariance: with Philox noise keyed by the global clip id, a clip's sample is
    bit-identical whether it is sampled alone, as part of a larger batch, or at another batch position - no kernel's
    summation order depends on the batch (no split-K, LayerNorm and attention are per row / per clip)."""
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd = O.det_state_dict(arch, tag="rob/w")
    B, T = 5, 24
    cond = _cond(B, T, "shardinv")
    ctx = _ctx(arch, sd, B, T, prec, n_steps=6)
    _set_cond(ctx, cond)
    full = ctx.sample_loop(noise=None, seed=11, clip_id_base=100).cpu()

    def sub(idx):
        c = {k: (v[idx] if isinstance(v, torch.Tensor) else [v[i] for i in idx]) for k, v in cond.items()}
        return c

    for idx in ([3], [1, 2], [4, 0]):
        # contiguous global ids only when the picked clips are contiguous: sample them one sub-batch per id base
        for j, i in enumerate(idx):
            _set_cond(ctx, sub([i]))
            one = ctx.sample_loop(noise=None, seed=11, clip_id_base=100 + i).cpu()
            assert torch.equal(one[0], full[i]), (prec, i)
    _set_cond(ctx, sub([1, 2]))
    two = ctx.sample_loop(noise=None, seed=11, clip_id_base=101).cpu()
    assert torch.equal(two, full[1:3])
    ctx.close()


def test_graph_is_captured_once_per_shape_and_spans_several_steps():
    """The loop graph holds the largest divisor of N up to 16 steps and is step-, seed-, clip-range- and noise-agnostic:
    new seeds / clip bases / supplied-noise tensors replay the same executable graph; only a new (B, T, N) re-captures."""
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd = O.det_state_dict(arch, tag="rob/w")
    ctx = _ctx(arch, sd, 3, 16, "f32", n_steps=1000)
    _set_cond(ctx, _cond(3, 16, "graph"))
    outs = [ctx.sample_loop(noise=None, seed=s, clip_id_base=b).cpu() for s, b in ((1, 0), (2, 0), (2, 7))]
    assert ctx.loop_stats() == (1, 100)  # one capture, 1000 / 10 launches per loop
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    again = ctx.sample_loop(noise=None, seed=1, clip_id_base=0).cpu()
    assert torch.equal(again, outs[0]) and ctx.loop_stats() == (1, 100)
    ctx.close()
    # step counts without a divisor in 2..16 fall back to one step per graph; N = 12 is one 12-step graph
    for N, launches in ((17, 17), (12, 1), (34, 17)):
        ctx = _ctx(arch, sd, 2, 16, "f32", n_steps=N)
        cond = _cond(2, 16, "graph")
        _set_cond(ctx, cond)
        g = torch.Generator().manual_seed(N)
        draws = torch.randn(N + 1, 2, 99, 1, 16, generator=g)
        out = ctx.sample_loop(noise=draws).cpu()
        ref = O.sample_loop(sd, arch, O.make_tables(N, "cosine"), cond, (2, 99, 1, 16), lambda k: draws[k])
        assert float((out - ref).abs().max()) < 1e-5, N
        assert ctx.loop_stats() == (1, launches), (N, ctx.loop_stats())
        # supplied noise of another loop: same graph
        out2 = ctx.sample_loop(noise=draws.flip(1).contiguous()).cpu()
        assert ctx.loop_stats()[0] == 1 and not torch.equal(out, out2)
        ctx.close()


# ---- range guard of the split-fp16 mode (include/tamf_hip.h: tamf_get_status_flags, TAMF_ERR_RANGE) --------------------------
def _blow_up_ffn_hidden(sd, scale):
    """linear1 of layer 0 scaled so that the FFN hidden activations H = GELU(W1 x + b1) leave the fp16 range (x is a
    LayerNorm output, O(1)): finite in fp32, not representable as split fp16."""
    sd = {k: v.clone() for k, v in sd.items()}
    sd["seqTransEncoder.layers.0.linear1.weight"] *= scale
    return sd


def test_f16x3_activation_overflow_raises_the_status_flag():
    from oakink2_tamf_amd.hip_backend import STATUS_F16_RANGE
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd = O.det_state_dict(arch, tag="rob/w")
    cond = _cond(2, 16, "range")
    x = torch.randn(2, 99, 1, 16, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([10, 700])
    # in range: no flag, in any call
    ctx = _ctx(arch, sd, 2, 16, "f16x3")
    _set_cond(ctx, cond)
    assert ctx.status_flags(clear=False) == 0  # the word belongs to this context and starts clear
    ctx.denoise(x, t)
    ctx.sample_loop(noise=None, seed=1)
    assert ctx.status_flags() == 0
    ctx.close()
    # W1 large but representable (max |w| ~ 3e4 < 65504): the weights load, the hidden activations overflow -> flag
    w1 = sd["seqTransEncoder.layers.0.linear1.weight"]
    scale = 3.0e4 / float(w1.abs().max())
    big = _blow_up_ffn_hidden(sd, scale)
    ref = O.denoiser_forward(big, arch, x, t, cond)
    assert torch.isfinite(ref).all()  # fp32 (the reference's arithmetic) is fine with these weights
    ctx = _ctx(arch, big, 2, 16, "f16x3")
    _set_cond(ctx, cond)
    ctx.denoise(x, t)
    assert ctx.status_flags(clear=False) & STATUS_F16_RANGE
    assert ctx.status_flags() & STATUS_F16_RANGE  # sticky until cleared
    assert ctx.status_flags() == 0
    ctx.close()
    # a raw context built with range_check=True raises by itself (no fallback at this level)
    from oakink2_tamf_amd.hip_backend import TamfContext, TamfRangeError
    from test_hip_forward import _arch_dict

    ctx = TamfContext(_arch_dict(arch), 2, 16, precision="f16x3", range_check=True)
    ctx.load_state_dict(big)
    tab = O.make_tables(1000, "cosine")
    ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
    _set_cond(ctx, cond)
    with pytest.raises(TamfRangeError):
        ctx.denoise(x, t)
    assert ctx.status_flags() == 0  # read and cleared by the raise
    ctx.close()
    # the same weights in f32: correct, and no flag
    ctx = _ctx(arch, big, 2, 16, "f32")
    _set_cond(ctx, cond)
    out = ctx.denoise(x, t).cpu()
    assert ctx.status_flags() == 0
    assert float((out - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max()))
    ctx.close()


def test_f16x3_weights_of_any_finite_magnitude_load_and_only_non_finite_ones_are_refused():
    """Round 4: split-fp16 weights are stored scaled by a per-tensor power of two (max |w| in [2^14, 2^15)) that the epilogue
    takes out again, so a weight beyond 65504 - or a whole tensor of tiny weights - is no reason to refuse a checkpoint; the
    result stays at the fp32 gate.  Only a non-finite weight is TAMF_ERR_RANGE."""
    from oakink2_tamf_amd.hip_backend import TamfContext, TamfRangeError
    from oracle import mdm_oracle as O
    from test_hip_forward import _arch_dict, _set_cond

    arch = O.ARCH_TINY
    base = O.det_state_dict(arch, tag="rob/w")
    cond = _cond(2, 16, "range")
    x = torch.randn(2, 99, 1, 16, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([10, 700])
    # (a) one huge weight in linear2 (its product goes to the fp32 LayerNorm, never into an fp16 operand)
    sd = {k: v.clone() for k, v in base.items()}
    sd["seqTransEncoder.layers.1.linear2.weight"][3, 7] = 7.0e4
    # (b) every encoder weight scaled down by 2^-12 with the LayerNorm after it undoing nothing: products of ~1e-5 - unscaled, the
    #     lo planes of such weights would be fp16 subnormals or zero (~8 significand bits left)
    sd_small = {k: v.clone() for k, v in base.items()}
    for k in sd_small:
        if k.endswith("linear1.weight") or k.endswith("in_proj_weight"):
            sd_small[k] *= 2.0 ** -12
    for tag, w in (("huge", sd), ("tiny", sd_small)):
        ref = O.denoiser_forward(w, arch, x, t, cond)
        assert torch.isfinite(ref).all()
        ctx = TamfContext(_arch_dict(arch), 2, 16, precision="f16x3")
        ctx.load_state_dict(w)
        _set_cond(ctx, cond)
        out = ctx.denoise(x, t).cpu()
        # round 5: one power of two per tensor - the 7e4 outlier leaves the ordinary weights of ITS tensor with fewer than 22
        # significand bits; that is reported (a status bit set at tamf_finalize_weights, the tensor named), not refused
        from oakink2_tamf_amd.hip_backend import STATUS_F16_WEIGHT_RANGE, lib

        assert ctx.status_flags() == (STATUS_F16_WEIGHT_RANGE if tag == "huge" else 0), tag
        if tag == "huge":
            assert b"layers.1.linear2.weight" in lib().tamf_last_error(ctx._h)
            ctx2 = TamfContext(_arch_dict(arch), 2, 16, precision="f16x3")
            with pytest.raises(TamfRangeError, match="layers.1.linear2.weight"):
                ctx2.load_state_dict(w, strict_weight_range=True)
            ctx2.close()
        err = float((out - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err < 1e-5, (tag, err)
        ctx.close()
    for bad in (float("inf"), float("nan")):
        sdn = {k: v.clone() for k, v in base.items()}
        sdn["seqTransEncoder.layers.1.linear2.weight"][3, 7] = bad
        ctx = TamfContext(_arch_dict(arch), 2, 16, precision="f16x3")
        with pytest.raises(TamfRangeError, match="layers.1.linear2.weight"):
            ctx.load_state_dict(sdn)
        ctx.close()
    for prec in ("bf16x3", "f32"):  # the other formats take any float32
        ctx = TamfContext(_arch_dict(arch), 2, 16, precision=prec)
        ctx.load_state_dict(sd)
        ctx.close()


def test_status_word_is_per_context():
    """Two live f16x3 contexts on one device: an overflow in one raises only ITS bit; reading / clearing one leaves the other's
    evidence alone; a context created while another's bit is set starts clear and clears nothing (ADVICE r3 / VERDICT r3 #3)."""
    from oakink2_tamf_amd.hip_backend import STATUS_F16_RANGE
    from oracle import mdm_oracle as O
    from test_hip_forward import _set_cond

    arch = O.ARCH_TINY
    sd = O.det_state_dict(arch, tag="rob/w")
    w1 = sd["seqTransEncoder.layers.0.linear1.weight"]
    big = _blow_up_ffn_hidden(sd, 3.0e4 / float(w1.abs().max()))
    cond = _cond(2, 16, "range")
    x = torch.randn(2, 99, 1, 16, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([10, 700])
    good = _ctx(arch, sd, 2, 16, "f16x3")
    bad = _ctx(arch, big, 2, 16, "f16x3")
    _set_cond(good, cond)
    _set_cond(bad, cond)
    ref = good.denoise(x, t).cpu()
    bad.denoise(x, t)
    out = good.denoise(x, t).cpu()  # interleaved with the overflowing context on the same device and stream
    assert good.status_flags(clear=True) == 0            # ... and this read-and-clear must not touch the other word
    assert bad.status_flags(clear=False) & STATUS_F16_RANGE
    third = _ctx(arch, sd, 2, 16, "f16x3")               # creating a context clears nobody's evidence
    assert third.status_flags(clear=False) == 0
    assert bad.status_flags(clear=True) & STATUS_F16_RANGE
    assert bad.status_flags() == 0 and good.status_flags() == 0
    assert torch.equal(out, ref)
    for c in (good, bad, third):
        c.close()


@pytest.mark.parametrize("how", ["activation", "weight", "nonfinite_weight"])
def test_module_falls_back_to_f32_when_the_fp16_range_is_left(how):
    """The drop-in module's default precision is f16x3 with range_check='fallback': out-of-range activations (caused by a scaled
    tensor or by one huge weight - the weight itself loads since round 4) or a non-finite weight make it repeat the call in f32
    - bit-identical to a module built with precision='f32' - and stay there; range_check='raise' raises instead."""
    from oakink2_tamf_amd.hip_backend import TamfRangeError
    from oakink2_tamf_amd.model.diffusion_util import create_gaussian_diffusion
    from oakink2_tamf_amd.model.interaction_segment_mdm import InterationSegmentMDM
    from oracle import mdm_oracle as O

    arch = O.ARCH_TINY
    kw = dict(latent_dim=arch.latent_dim, ff_size=arch.ff_size, num_layers=arch.num_layers, num_heads=arch.num_heads)
    sd = O.det_state_dict(arch, tag="rob/w")
    if how == "activation":
        w1 = sd["seqTransEncoder.layers.0.linear1.weight"]
        sd = _blow_up_ffn_hidden(sd, 3.0e4 / float(w1.abs().max()))
    elif how == "weight":  # loads (pre-scaled), but the hidden activations it produces leave the fp16 range
        sd = {k: v.clone() for k, v in sd.items()}
        sd["seqTransEncoder.layers.0.linear1.weight"][:, 0] = 1.0e6
    else:  # refused at load -> f32 from the start; fp32 turns the inf into NaN and nan_to_num zeroes it: both modules agree
        sd = {k: v.clone() for k, v in sd.items()}
        sd["seqTransEncoder.layers.0.linear1.weight"][0, 0] = float("inf")
    cond = _cond(2, 16, "range")
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in cond.items()}
    x = torch.randn(2, 99, 1, 16, generator=torch.Generator().manual_seed(5)).cuda()
    t = torch.tensor([10, 700]).cuda()
    m = InterationSegmentMDM(**kw).cuda()
    assert m.precision == "f16x3" and m.range_check == "fallback"
    m.load_state_dict(sd)
    m32 = InterationSegmentMDM(**kw, precision="f32").cuda()
    m32.load_state_dict(sd)
    out, ref = m(x, t, batch), m32(x, t, batch)
    assert m.active_precision == "f32" and torch.equal(out, ref) and torch.isfinite(ref).all()
    dif = create_gaussian_diffusion(diffusion_steps=6, noise_schedule="cosine")
    m.load_state_dict(sd)  # new weights: the requested arithmetic is tried again
    assert m.active_precision == "f16x3"
    s1 = dif.p_sample_loop(m, (2, 99, 1, 16), clip_denoised=False, model_kwargs={"batch": batch}, seed=3)
    s2 = dif.p_sample_loop(m32, (2, 99, 1, 16), clip_denoised=False, model_kwargs={"batch": batch}, seed=3)
    assert m.active_precision == "f32" and torch.equal(s1, s2)
    mr = InterationSegmentMDM(**kw, range_check="raise").cuda()
    mr.load_state_dict(sd)
    with pytest.raises(TamfRangeError):
        mr(x, t, batch)
