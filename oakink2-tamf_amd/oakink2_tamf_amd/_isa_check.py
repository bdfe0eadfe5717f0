"""Build-time check of the counted waits of the clip-tile GEMM (csrc/tamf_gemm_clip.h); run by _lib.build on the device assembly of
the very compile that produced the library (hipcc -save-temps), and by tools/check_clip_stores.py / tests/test_isa_clip_waits.py.

The loader waves of clip_gemm_kernel guard LDS reuse with `s_waitcnt vmcnt(PH + SX)`: SX is the number of global-store
instructions the epilogue of a tile issues behind the LDS-DMA requests of the next K tile.  That number is a compile-time formula
(MSUBX * NCHUNK * CHUNK_STORES, or the V^T form), so the kernel is only correct while hipcc emits exactly ONE store instruction per
source-level 16-byte store - no merging, splitting or elision.  check(path) reads the gfx950 assembly and verifies, for every
clip_gemm_kernel instantiation:
  * the static number of global_store instructions equals what the source's formula implies for all its code variants
    (three activation variants x the X and Y wave roles; V^T form: one variant per role);
  * every counted wait immediate as large as the kernel's own (s_waitcnt vmcnt(N), N >= min(PH - 1, SX); the compiler's small
    counted waits for the column constants lie below) is one of {PH, PH - 1, PH + SX, PH - 1 + SX, SX};
  * no buffer_store / flat_store / scratch instruction appears (a spill or another store flavour would not be counted by the formula).
A different hipcc may merge or split stores; _lib.build then rebuilds with -DTAMF_CLIP_SAFE_WAIT (every counted wait = vmcnt(0))."""
import re

OPS = {"OpF32": (0, False), "OpBF16": (1, False), "OpBF16X3": (2, True), "OpF16X3": (3, True)}  # name -> (PREC, SPLIT)


class IsaMismatch(RuntimeError):
    pass

def kernels(path):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z16clip_gemm_kernelI\w+):", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            if j >= len(lines):
                raise IsaMismatch("no .Lfunc_end label behind " + m.group(1) + ": unknown assembly layout")
            yield m.group(1), lines[i:j]
            i = j
        i += 1


def parse(name):
    m = re.match(r"_Z16clip_gemm_kernelI\d+(Op[A-Z0-9]+)Li(\d+)ELi(\d+)ELi(\d+)E\d+(Epi[A-Za-z0-9]+?)(?:IS0_(?:Lb[01]E)?E)?Ev", name)
    if not m:
        raise IsaMismatch("cannot parse " + name)
    return m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5)


def expected(op, nsub, ni, xsub, epi):
    prec, split = OPS[op]
    msx, msy = xsub, nsub - xsub
    npiece = (nsub * 16 + 64 * ni) // 8
    ph = (npiece + 3) // 4
    if epi == "EpiVt":
        cs = 2 if split else 1
        per = (lambda ms: ms if prec == 0 else (ms + 1) // 2)
        sx = per(msx) * ni * cs
        static = (per(msx) + per(msy)) * ni * cs
    else:
        hook = 0
        if epi == "EpiStoreF32":
            ch, cs = 4, 1
        elif epi == "EpiSeqRows":  # fp32 row pieces + operand pieces; + the Y waves' prefix-row hook (one loop body: fp32 2 + operand)
            ch = 4 if prec == 0 else 8
            cs = ch // 4 + (2 if split else 1)
            hook = 2 + (2 if (split or prec == 0) else 1)
        elif epi == "EpiResid":  # deferred LayerNorm: fp32 row piece (2 x 16 bytes) + operand piece(s) (f32: none, the row piece is the operand) + the block statistics (8 bytes)
            ch, cs = 8, 2 + (0 if prec == 0 else 2 if split else 1) + 1
        else:  # EpiBiasAct (with or without the deferred LayerNorm of its rows), EpiQK: operand output
            ch, cs = (4 if prec == 0 else 8), (2 if split else 1)
        nchunk = 4 * ni // ch
        sx = msx * nchunk * cs
        # three activation variants per wave role: GELU and SiLU run their row tiles in a loop (one row tile of stores in the text),
        # the plain one is unrolled (MS row tiles)
        static = nchunk * cs * ((2 + msx) + (2 + msy)) + hook
    return ph, sx, static


def check(path, log=None):
    """Returns the number of instantiations checked; raises IsaMismatch with the per-kernel report when one does not match."""
    bad = 0
    n = 0
    report = []
    for name, body in kernels(path):
        op, nsub, ni, xsub, epi = parse(name)
        ph, sx, static = expected(op, nsub, ni, xsub, epi)
        ops = [l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        stores = sum(1 for o in ops if o.startswith("global_store"))
        other = sorted({o for o in ops if o.startswith(("buffer_store", "flat_store", "scratch_"))})
        allowed = {ph, ph - 1, ph + sx, ph - 1 + sx, sx}
        waits = sorted({int(m.group(1)) for l in body for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m and int(m.group(1)) >= min(ph - 1, sx)})
        ok = stores == static and not other and all(w in allowed for w in waits)
        n += 1
        line = (f"{'ok ' if ok else 'BAD'} {op:8s} NSUB={nsub:2d} NI={ni} XSUB={xsub} {epi:12s} PH={ph:2d} SX={sx:2d}: global_store {stores} (expected {static}), "
                f"counted waits {waits} (allowed {sorted(allowed)}){' other stores: ' + ','.join(other) if other else ''}")
        report.append(line)
        if log:
            log(line)
        bad += 0 if ok else 1
    if n == 0:
        raise IsaMismatch("no clip_gemm_kernel found in " + path)
    if bad:
        raise IsaMismatch(f"{bad} of {n} clip_gemm_kernel instantiations do not match the store-count formula of tamf_gemm_clip.h:\n"
                          + "\n".join(l for l in report if l.startswith("BAD")))
    return n


def check_deep(path, log=None):
    """The small tiles with the deep K pipeline (csrc/tamf_gemm_deep.h): a wave waits with s_waitcnt vmcnt(PW * (NSTG - 2)) for the K tile
    it reads next, PW = the LDS-DMA pieces it requests per K tile - correct only while hipcc emits exactly one global_load_lds per source-level
    request (NSTG - 1 batches in the prologue, one in the K loop) and keeps the two counted waits.  Returns the number of instantiations."""
    lines = open(path).read().split("\n")
    n, bad, i = 0, [], 0
    while i < len(lines):
        m = re.match(r"^(_Z16gemm_deep_kernelI\d+(Op[A-Z0-9]+)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\w+):", lines[i])
        if m:
            j = i
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            if j >= len(lines):
                raise IsaMismatch("no .Lfunc_end label behind " + m.group(1))
            body = lines[i:j]
            bm, bn, wgm, wgn, nstg = (int(m.group(k)) for k in range(3, 8))
            nwv, a_pieces, w_pieces = wgm * wgn, bm // 8, bn // 8
            a_pw, w_pw, a_rem = -(-a_pieces // nwv), w_pieces // nwv, a_pieces % nwv
            want_dma = nstg * (a_pw + w_pw)
            waits = {int(x.group(1)) for l in body for x in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if x}
            need = {(a_pw + w_pw) * (nstg - 2)} | ({(a_pw - 1 + w_pw) * (nstg - 2)} if a_rem else set())
            dma = sum(1 for l in body if l.strip().startswith("global_load_lds"))
            spill = any(l.strip().startswith("scratch_") for l in body)
            # the prologue's LDS-DMA stages must have landed at the first barrier: an s_waitcnt that waits for vmcnt(0) between the last
            # prologue request (the (NSTG - 1)-th batch) and the first s_barrier
            first_bar = next((k for k, l in enumerate(body) if l.strip().startswith("s_barrier")), len(body))
            dma_idx = [k for k, l in enumerate(body) if l.strip().startswith("global_load_lds")]
            pro_last = dma_idx[(nstg - 1) * (a_pw + w_pw) - 1] if len(dma_idx) >= (nstg - 1) * (a_pw + w_pw) else -1
            pro_wait = pro_last >= 0 and pro_last < first_bar and any(
                re.search(r"s_waitcnt\b.*vmcnt\(0\)", l) for l in body[pro_last:first_bar])
            ok = dma == want_dma and need <= waits and not spill and pro_wait
            line = (f"{'ok ' if ok else 'BAD'} {m.group(2):8s} {bm}x{bn} NSTG={nstg}: global_load_lds {dma} (expected {want_dma}), counted waits needed {sorted(need)}, "
                    f"seen {sorted(waits)}, vmcnt(0) between the prologue's requests and the first barrier: {pro_wait}")
            if log:
                log(line)
            if not ok:
                bad.append(line)
            n += 1
            i = j
        i += 1
    if bad:
        raise IsaMismatch(f"{len(bad)} of {n} gemm_deep_kernel instantiations do not match the request counts of tamf_gemm_deep.h:\n" + "\n".join(bad))
    return n


SCRATCH_LIMIT = 32  # bytes per lane


def scratch_report(path, prefixes=("_Z15attn_res_kernel", "_Z11attn_kernel", "_Z16clip_gemm_kernel")):
    """[(kernel, private segment bytes, vgprs)] of the hot kernels, from their .amdhsa_kernel descriptors.  Scratch in these kernels is
    register spilling (none of them indexes a private array at run time): round 4 shipped short-clip attention instantiations with
    880 - 1 048 bytes of it per lane, unnoticed."""
    text = open(path).read()
    out = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if not name.startswith(prefixes):
            continue
        ps = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        out.append((name, int(ps.group(1)) if ps else 0, int(vg.group(1)) if vg else -1))
    return out


def check_scratch(path, limit=SCRATCH_LIMIT, log=None):
    """Raises IsaMismatch when an attention / clip-GEMM kernel keeps more than `limit` bytes of scratch per lane; returns the number of
    kernels looked at."""
    rep = scratch_report(path)
    bad = [(n, ps, vg) for n, ps, vg in rep if ps > limit]
    if log:
        for n, ps, vg in rep:
            if ps:
                log(f"scratch {ps:5d} B  vgprs {vg:3d}  {n}")
    if not rep:
        raise IsaMismatch("no attention / clip-GEMM kernel descriptor found in " + path)
    if bad:
        raise IsaMismatch(f"{len(bad)} hot kernel(s) spill more than {limit} bytes per lane:\n" + "\n".join(f"  {ps} B ({vg} vgprs) {n}" for n, ps, vg in bad))
    return len(rep)
