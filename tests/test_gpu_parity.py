"""Parity tests proper: HIP path (through the C ABI) vs the oracle / the real reference.

Bars: byte-identical to liblzma 5.8.3's MT encoder for presets 0-3 when one span covers a Block;
byte-identical to the oracle's span-mode restatement otherwise; bit-exact round trip always."""
import hashlib
import os

import numpy as np
import pytest

import _oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc():
    import torch
    import xz_amd
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    e = xz_amd.Encoder(0)
    yield e
    e.close()


def gpu_encode(enc, data, opts, block_size):
    import torch
    t = (torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda() if len(data)
         else torch.empty(0, dtype=torch.uint8, device="cuda"))
    out, binfo = enc.encode(t, opts=opts, block_size=block_size)
    return out.cpu().numpy().tobytes(), binfo


def inputs():
    rng = np.random.default_rng(21)
    lorem = o.corpus_lorem(1 << 20)
    rnd = bytes(rng.integers(0, 256, size=300000, dtype=np.uint8))
    return {
        "one": b"x", "two": b"xy", "three": b"aaa", "four": b"abcd", "five": b"abcab",
        "run": b"\0" * 100000, "period3": b"abc" * 30000, "text229001": lorem[:229001],
        "lorem1M": lorem, "mixed": o.corpus_mixed(700000, 5), "rnd": rnd,
        "sandwich": lorem[:150000] + rnd[:200000] + lorem[:100000],
        "long_match": lorem[:5000] * 40,
        "abc": o.corpus_abc(), "random_lcg": o.corpus_random(),
    }


@pytest.mark.parametrize("preset", [0, 1, 2, 3])
def test_identical_to_reference_whole_block_spans(enc, preset):
    import xz_amd
    opts = xz_amd.preset_options(preset, span_size=xz_amd.SPAN_WHOLE_BLOCK)
    prm = o.params_for_gpu_options(opts)
    for name, data in inputs().items():
        for bs in (1 << 20, 200000):
            got, _ = gpu_encode(enc, data, opts, bs)
            want = o.orc_xz_stream(data, prm, bs)
            assert o.first_diff(got, want) == -1, (name, preset, bs)
            if o.have_ref():
                ref = o.ref_encode_mt(data, preset, threads=2, block_size=bs)
                assert o.first_diff(got, ref) == -1, ("vs liblzma", name, preset, bs)


@pytest.mark.parametrize("preset,span", [(1, 4096), (1, 65536), (3, 16384), (6, 0), (9, 0), (0, 8192)])
def test_span_mode_identical_to_oracle(enc, preset, span):
    import xz_amd
    opts = xz_amd.preset_options(preset, span_size=span)
    prm = o.params_for_gpu_options(opts)
    for name, data in inputs().items():
        bs = 1 << 20
        got, binfo = gpu_encode(enc, data, opts, bs)
        want = o.orc_xz_stream(data, prm, bs)
        assert o.first_diff(got, want) == -1, (name, preset, span)
        r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
        assert r == 0 and dec == bytes(data)
        if o.have_ref():
            rr, rdec = o.ref_decode(got, len(data) + 16)
            assert rr == 1 and rdec == bytes(data), ("liblzma decoder", name)


@pytest.mark.parametrize("preset,parser,window,span", [
    (2, 1, None, 65536),      # exact HC4 finder with the windowed optimal parser
    (1, 1, None, 0xFFFFFFFF),
    (6, None, None, 0xFFFFFFFF),
    (6, None, 2, 8192),       # narrow suffix-order window
    (4, None, None, 0), (5, None, None, 4096), (7, None, None, 0),
    (3 | 0x80000000, None, None, 0),
])
def test_successor_finder_and_optimal_parser_identical_to_oracle(enc, preset, parser, window, span):
    """Presets 4-9 / -e (BT4 + normal mode in the reference) run OUR finder and parser; the bar is
    bit-exactness with their CPU restatement (oracle build_sa / find_sn / optimum_window) plus a
    bit-exact round trip through the real reference decoder."""
    import xz_amd
    opts = xz_amd.preset_options(preset, span_size=span)
    if parser is not None:
        opts.gpu_parser = parser
    if window is not None:
        opts.gpu_sa_window = window
    prm = o.params_for_gpu_options(opts)
    for name, data in inputs().items():
        if len(data) > 420000:
            data = data[:420000]
        got, _ = gpu_encode(enc, data, opts, 1 << 20)
        want = o.orc_xz_stream(data, prm, 1 << 20)
        assert o.first_diff(got, want) == -1, (name, preset, parser, span)
        if o.have_ref():
            rr, rdec = o.ref_decode(got, len(data) + 16)
            assert rr == 1 and rdec == bytes(data), ("liblzma decoder", name)


def test_suffix_order_and_match_lists_identical_to_oracle(enc):
    """The two device structures behind presets 4-9, stage by stage: the 32 / 64 / 128 / 256-byte-prefix suffix order of every
    Block (rocprim radix sorts + rank doubling vs the oracle's build_sa) and the per-position match-list
    records of k_find_sn (vs the oracle's find_sn), several Blocks per batch, ragged last Block."""
    import xz_amd
    cases = {"lorem": o.corpus_lorem(300000), "mixed": o.corpus_mixed(260000, 11),
             "runs": (b"a" * 70000 + b"ab" * 30000 + bytes(range(256)) * 300)[:200000],
             "text": xz_amd.corpus_text(250000, seed=3).tobytes()}
    for name, data in cases.items():
        for bs, span, window, depth in ((1 << 20, 0, None, None), (65536, 8192, None, 32), (100000, 0xFFFFFFFF, 3, 256),
                                        (1 << 20, 16384, None, 128)):
            opts = xz_amd.preset_options(6, span_size=span)
            if window is not None:
                opts.gpu_sa_window = window
            if depth is not None:
                opts.gpu_sa_depth = depth
            prm = o.params_for_gpu_options(opts)
            gpu_encode(enc, data, opts, bs)
            n = len(data)
            sa = enc.debug_fetch(1, n)
            rk = enc.debug_fetch(2, n)
            lists = enc.debug_fetch(3, 8 * n).reshape(n, 8)
            for b0 in range(0, n, bs):
                blk = data[b0:b0 + bs]
                osa, ork = o.orc_sa_dump(blk, opts.gpu_sa_depth)
                assert (sa[b0:b0 + len(blk)] == osa + b0).all(), ("suffix order", name, bs, b0,
                                                                  int(np.nonzero(sa[b0:b0 + len(blk)] != osa + b0)[0][0]))
                assert (rk[b0:b0 + len(blk)] == ork + b0).all(), ("rank", name, bs, b0)
                want = o.orc_list_dump(blk, prm)
                got = lists[b0:b0 + len(blk)].copy()
                cnt = got[:, 7] & 0xFF
                for k in range(7):
                    got[cnt <= k, k] = 0            # entries past the count are undefined on the device
                bad = np.nonzero((got != want).any(axis=1))[0]
                bad = bad[bad > 0]                  # position 0 of a Block: no earlier data, never read by the parser
                assert len(bad) == 0, ("match lists", name, bs, b0, int(bad[0]), got[bad[0]].tolist(), want[bad[0]].tolist())


@pytest.mark.parametrize("n", [1, 5, 1000, 32767, 32768, 32769, 70001, 1200000])
def test_structure_build_edge_sizes(enc, n):
    """Sizes around the 32 Ki-position buckets of the by-position inversion (and below one bucket, where no radix
    pass runs at all), Blocks smaller and larger than the input, more than 256 Blocks (one sort of everything plus a
    sort by Block number instead of one sort per Block): suffix order, rank and whole Stream vs the oracle."""
    import xz_amd
    data = xz_amd.corpus_text(max(n, 4096), seed=17).tobytes()[:n]
    for bs in (4096, 1 << 20):
        opts = xz_amd.preset_options(6)
        prm = o.params_for_gpu_options(opts)
        got, _ = gpu_encode(enc, data, opts, bs)
        assert o.first_diff(got, o.orc_xz_stream(data, prm, bs)) == -1, (n, bs)
        sa = enc.debug_fetch(1, n)
        rk = enc.debug_fetch(2, n)
        for b0 in range(0, n, bs):
            blk = data[b0:b0 + bs]
            osa, ork = o.orc_sa_dump(blk, opts.gpu_sa_depth)
            assert (sa[b0:b0 + len(blk)] == osa + b0).all(), ("suffix order", n, bs, b0)
            assert (rk[b0:b0 + len(blk)] == ork + b0).all(), ("rank", n, bs, b0)


def test_span_plan_identical_to_oracle(enc):
    """Cost-balanced spans, stage by stage: the per-chunk work / bit estimates of k_span_est and the cuts of k_span_cut
    vs the oracle's est_chunk / plan_spans (several Blocks per batch, a ragged last Block, text next to long runs
    and incompressible bytes so that the spans differ in length a lot), then the whole Stream."""
    import xz_amd
    rng = np.random.default_rng(5)
    lorem = o.corpus_lorem(600000)
    mix = (lorem[:300000] + b"\0" * 400000 + bytes(rng.integers(0, 256, size=200000, dtype=np.uint8))
           + (lorem[:4000] * 200) + xz_amd.corpus_text(500000, seed=9).tobytes())
    import _corpora
    sparse = _corpora.sparse_text(5 << 20)        # < 0.5 estimated bits per byte: the bit bound of a piece scales up (plan_spans_ex)
    for bs, cost, bits in ((1 << 20, 0, None), (700000, 40000, 50000), (1 << 21, 20000, 0), (3 << 20, 0, None), (3 << 20, 0, 30000)):
        if bs == 3 << 20:
            mix = sparse
        opts = xz_amd.preset_options(6)
        if cost:
            opts.span_cost = cost
        if bits is not None:
            opts.span_bits = bits
        got, _ = gpu_encode(enc, mix, opts, bs)
        st = enc.stats()
        assert st.span_size == 0 and st.span_cost_used == opts.span_cost
        prm = o.params_for_gpu_options(opts, span_cost_used=st.span_cost_used)
        nb = (len(mix) + bs - 1) // bs
        spb, cpb = bs // 65536 + 2, (bs + 4095) // 4096
        tab = enc.debug_fetch(5, 2 * nb * spb).reshape(nb, spb, 2)
        cnt = enc.debug_fetch(6, nb)
        est = enc.debug_fetch(7, 2 * nb * cpb).reshape(2, nb, cpb)
        total_spans = 0
        for b in range(nb):
            blk = mix[b * bs:(b + 1) * bs]
            work, bitsv, starts = o.orc_span_plan(blk, prm)
            m = len(work)
            assert (est[0, b, :m] == work).all(), ("work estimates", bs, b, int(np.nonzero(est[0, b, :m] != work)[0][0]))
            assert (est[1, b, :m] == bitsv).all(), ("bit estimates", bs, b)
            assert cnt[b] == len(starts), ("span count", bs, b, int(cnt[b]), len(starts))
            assert (tab[b, :cnt[b], 0] == starts + b * bs).all(), ("span starts", bs, b)
            ends = np.append(starts[1:], len(blk)) + b * bs
            assert (tab[b, :cnt[b], 1] == ends).all(), ("span ends", bs, b)
            total_spans += len(starts)
        assert st.spans == total_spans
        assert o.first_diff(got, o.orc_xz_stream(mix, prm, bs)) == -1, ("stream", bs, cost)
        rr, rdec = o.ref_decode(got, len(mix) + 16)
        assert rr == 1 and rdec == mix


def _lzma_code_encode(L, data, preset, block_size, piece):
    """Encode through lzma_stream_encoder_mt / lzma_code, feeding `piece` bytes per LZMA_RUN call, then LZMA_FINISH."""
    import ctypes as C
    import sys
    sys_path = os.path.join(o.ROOT, "tools")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from bench_lzma_code import Mt, Stream
    ib = C.create_string_buffer(data, len(data))
    ob = C.create_string_buffer(len(data) + (1 << 20))
    s = Stream()
    m = Mt(threads=1, preset=preset, check=4, block_size=block_size, timeout=0)
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == 0
    s.next_out = C.cast(ob, C.c_void_p).value
    s.avail_out = len(ob)
    pos = 0
    while pos < len(data):
        k = min(piece, len(data) - pos)
        s.next_in = C.cast(ib, C.c_void_p).value + pos
        s.avail_in = k
        while s.avail_in:
            assert L.lzma_code(C.byref(s), 0) == 0
        pos += k
    r = L.lzma_code(C.byref(s), 3)
    while r == 0:
        r = L.lzma_code(C.byref(s), 3)
    assert r == 1
    out = ob.raw[: s.total_out]
    L.lzma_end(C.byref(s))
    return out


def _walk_symbols(sl, sd, gsl, gsd, n):
    """First symbol start at which the recorded parses differ (None: identical)."""
    p = 0
    while p < n:
        if sl[p] != gsl[p] or sd[p] != gsd[p]:
            return p
        p += max(1, int(sl[p]) & 0x7FFF)           # bit 15 = "coded as a match" flag
    return None


@pytest.mark.parametrize("preset,single_phase", [(6, False), (4, False), (9 | 0x80000000, False), (6, True)])
def test_two_phase_stages_identical_to_oracle(enc, preset, single_phase):
    """The two-phase path stage by stage (oracle: plan_spans_ex / parse_block / encode_syms): the parse pieces incl. the
    64 KiB seed piece, the encode spans cut at piece ends, the recorded symbols of every piece (seed model as prior) and
    the bytes; several Blocks per batch, a ragged last Block, enc_span_bits small enough for several encode spans per
    Block.  single_phase = the same options with enc_span_bits = 0 (every span of the plan resets the coder state)."""
    import xz_amd
    rng = np.random.default_rng(7)
    lorem = o.corpus_lorem(900000)
    mix = (xz_amd.corpus_text(1500000, seed=3).tobytes() + lorem[:700000] + b"\0" * 300000
           + bytes(rng.integers(0, 256, size=150000, dtype=np.uint8)) + (lorem[:3000] * 150) + xz_amd.corpus_tar(1200000).tobytes())
    bs = 1 << 21
    opts = xz_amd.preset_options(preset)
    opts.span_cost = 50000
    opts.span_bits = 0
    opts.enc_span_bits = 0 if single_phase else 300000
    got, _ = gpu_encode(enc, mix, opts, bs)
    st = enc.stats()
    prm = o.params_for_gpu_options(opts)
    assert prm.enc_bits == opts.enc_span_bits
    nb = (len(mix) + bs - 1) // bs
    if single_phase:
        assert o.first_diff(got, o.orc_xz_stream(mix, prm, bs)) == -1
        rr, rdec = o.ref_decode(got, len(mix) + 16)
        assert rr == 1 and rdec == mix
        assert st.enc_spans == 0
        return
    spb, esb = bs // 65536 + 2, bs // (256 << 10) + 1
    tab = enc.debug_fetch(5, 2 * nb * spb).reshape(nb, spb, 2)
    cnt = enc.debug_fetch(6, nb)
    etab = enc.debug_fetch(11, 2 * nb * esb).reshape(nb, esb, 2)
    ecnt = enc.debug_fetch(12, nb)
    gsl = enc.debug_fetch(9, len(mix), "uint16")
    gsd = enc.debug_fetch(10, len(mix), "uint32")
    # round 6: what the carried model walk over iteration 1's records hands every piece (state, rep distances; the model itself
    # is overwritten by the piece's own adaptation in iteration 2), the raw decision per piece, the carry decision per encode span
    gsr = enc.debug_fetch(14, 8 * nb * spb).reshape(nb, spb, 8)
    gpi = enc.debug_fetch(13, 16 * nb * spb).reshape(nb, spb, 16)
    gcarry = enc.debug_fetch(16, nb * esb).reshape(nb, esb)
    pieces = spans = carried = stored = 0
    for b in range(nb):
        blk = mix[b * bs:(b + 1) * bs]
        starts, estarts = o.orc_piece_plan(blk, prm)
        assert cnt[b] == len(starts) and (tab[b, :cnt[b], 0] == starts + b * bs).all(), ("pieces", b)
        assert int((tab[b, :cnt[b], 1] - tab[b, :cnt[b], 0]).max()) <= 1 << 20          # no piece longer than 1 MiB (round 6)
        if len(blk) > 65536:
            assert starts[1] == 65536                                   # the seed piece
        assert ecnt[b] == len(estarts) and (etab[b, :ecnt[b], 0] == estarts + b * bs).all(), ("encode spans", b)
        assert set(estarts) <= set(starts)                               # encode spans end at piece ends
        assert (np.diff(estarts) >= (256 << 10)).all()
        eends = np.append(estarts[1:], len(blk)) + b * bs
        assert (etab[b, :ecnt[b], 1] == eends).all()
        osr, _, oprice, ocarry = o.orc_two_phase_debug(blk, prm, len(starts), len(estarts))
        assert (gsr[b, 1:cnt[b], :5] == osr[1:]).all(), ("state / rep distances at the piece starts", b,
                                                         np.nonzero((gsr[b, 1:cnt[b], :5] != osr[1:]).any(axis=1))[0][:5])
        plen = np.diff(np.append(starts, len(blk)))
        oraw = (plen >= 32768) & (oprice // 128 >= plen)
        assert (gpi[b, :cnt[b], 8 + 5] == oraw).all(), ("stored pieces", b)
        stored += int(oraw.sum())
        assert (gcarry[b, 1:ecnt[b]] == ocarry[1:]).all(), ("carry decisions", b, gcarry[b, :ecnt[b]], ocarry)
        carried += int((ocarry[1:] == 1).sum())
        sl, sd = o.orc_parse_dump(blk, prm)
        bad = _walk_symbols(sl, sd, gsl[b * bs:b * bs + len(blk)], gsd[b * bs:b * bs + len(blk)], len(blk))
        assert bad is None, ("symbol records", b, bad)
        pieces += len(starts)
        spans += len(estarts)
    assert st.spans == pieces and st.enc_spans == spans and spans > nb
    assert carried > 0 and stored > 0          # (the 150,000 random bytes of `mix`: a piece whose price says it does not shrink)
    assert o.first_diff(got, o.orc_xz_stream(mix, prm, bs)) == -1
    rr, rdec = o.ref_decode(got, len(mix) + 16)
    assert rr == 1 and rdec == mix


@pytest.mark.parametrize("part_iters", [2, 4])
def test_two_phase_part_iters_identical_to_oracle(enc, part_iters):
    """xzamd_lzma_options.part_iters: several PARTIAL parse iterations in front of the full one (each from the snapshots of the
    carried model walk over the records of the one before): byte-identical to the oracle, decodable by liblzma, and no larger
    than with one (a table of records "constants + counter + enum" is where the iterations pay: DESIGN.md 3.4b)."""
    import xz_amd
    import _corpora
    data = _corpora.random_class(10, 3 << 20) + xz_amd.corpus_text(2 << 20, seed=5).tobytes()
    bs = 1 << 22
    sizes = {}
    for it in (1, part_iters):
        opts = xz_amd.preset_options(6)
        opts.span_cost = 50000
        opts.span_bits = 0
        opts.enc_span_bits = 300000
        opts.part_iters = it
        got, _ = gpu_encode(enc, data, opts, bs)
        prm = o.params_for_gpu_options(opts)
        assert prm.part_iters == it
        assert o.first_diff(got, o.orc_xz_stream(data, prm, bs)) == -1, it
        rr, rdec = o.ref_decode(got, len(data) + 16)
        assert rr == 1 and rdec == data
        sizes[it] = len(got)
    assert sizes[part_iters] <= sizes[1] * 1.002, sizes


@pytest.mark.parametrize("pb,lc,lp", [(3, 3, 0), (4, 3, 0), (4, 0, 2), (2, 4, 0), (0, 1, 3)])      # (lc + lp = 4: the largest model, 14,134 probabilities)
def test_two_phase_pb_above_two_identical_to_oracle(enc, pb, lc, lp):
    """pb = 3, 4 with the optimal parser (lzma/lzma_common.h:32-37; refused until round 5): the parse pieces price with
    a pb = 2 view of the positions, the coder's continuous model runs the real pb (oracle: parse_block / encode_syms).
    Recorded symbols and bytes equal the oracle's, the Stream decodes through the reference decoder; the single-phase
    span kernel (one model for parser and coder) still refuses the option set."""
    import xz_amd
    import _corpora
    rng = np.random.default_rng(21)
    k = 60000
    rec = np.zeros((k, 16), dtype=np.uint8)                       # 16-byte records: what pb = 4 is for
    rec[:, 0:4] = np.arange(k, dtype=np.uint32).view(np.uint8).reshape(k, 4)
    rec[:, 4:8] = (np.sin(np.arange(k) * 7e-4) * 50).astype(np.float32).view(np.uint8).reshape(k, 4)
    rec[:, 8:16] = rng.integers(0, 4, size=(k, 8), dtype=np.uint8)
    mix = rec.tobytes() + xz_amd.corpus_text(900000, seed=4).tobytes() + _corpora.f64_sine(700000) + o.corpus_lorem(300000)
    bs = 1 << 21
    opts = xz_amd.preset_options(6)
    opts.pb, opts.lc, opts.lp = pb, lc, lp
    opts.span_cost = 50000
    opts.span_bits = 0
    opts.enc_span_bits = 300000
    got, _ = gpu_encode(enc, mix, opts, bs)
    prm = o.params_for_gpu_options(opts)
    assert prm.pb == pb and prm.enc_bits
    assert o.first_diff(got, o.orc_xz_stream(mix, prm, bs)) == -1
    rr, rdec = o.ref_decode(got, len(mix) + 16)
    assert rr == 1 and rdec == mix
    gsl = enc.debug_fetch(9, len(mix), "uint16")
    gsd = enc.debug_fetch(10, len(mix), "uint32")
    for b in range((len(mix) + bs - 1) // bs):
        blk = mix[b * bs:(b + 1) * bs]
        sl, sd = o.orc_parse_dump(blk, prm)
        bad = _walk_symbols(sl, sd, gsl[b * bs:b * bs + len(blk)], gsd[b * bs:b * bs + len(blk)], len(blk))
        assert bad is None, ("symbol records", b, bad)
    # default spans of the product (work target 131072, 0.8 Mbit encode spans)
    opts = xz_amd.preset_options(6)
    opts.pb, opts.lc, opts.lp = pb, lc, lp
    got, _ = gpu_encode(enc, mix, opts, bs)
    assert o.first_diff(got, o.orc_xz_stream(mix, o.params_for_gpu_options(opts, span_cost_used=enc.stats().span_cost_used), bs)) == -1
    rr, rdec = o.ref_decode(got, len(mix) + 16)
    assert rr == 1 and rdec == mix
    if pb > 2:
        opts.enc_span_bits = 0                                      # single phase: one model for parser and coder
        with pytest.raises(Exception):
            gpu_encode(enc, mix[:200000], opts, bs)


def test_two_phase_token_budget_overflow_identical_to_oracle(enc, monkeypatch):
    """Out of tokens (round-4 advisor, medium: far three-byte matches need more than the 10 tokens per byte the buffer
    holds) the model pass closes the chunk where it stands and stores the rest of the encode span raw.  Reached on
    ordinary data by turning the budget down on both sides; bytes identical to the oracle, decodable by liblzma."""
    import xz_amd
    rng = np.random.default_rng(5)
    data = (xz_amd.corpus_text(5 << 20, seed=9).tobytes() + bytes(rng.integers(0, 256, size=200000, dtype=np.uint8))
            + o.corpus_lorem(3 << 20))
    bs = 3 << 20
    opts = xz_amd.preset_options(6)
    opts.enc_span_bits = 400000
    prm = o.params_for_gpu_options(opts)
    full, _ = gpu_encode(enc, data, opts, bs)
    try:
        for budget in (1, 3):
            monkeypatch.setenv("XZAMD_TEST_TOK_PER_BYTE", str(budget))
            o.orc_set_tok_per_byte(budget)
            got, _ = gpu_encode(enc, data, opts, bs)
            assert o.first_diff(got, o.orc_xz_stream(data, prm, bs)) == -1, budget
            rr, rdec = o.ref_decode(got, len(data) + 16)
            assert rr == 1 and rdec == data
            assert len(got) > len(full)
    finally:
        o.orc_set_tok_per_byte(0)
        monkeypatch.delenv("XZAMD_TEST_TOK_PER_BYTE", raising=False)
    again, _ = gpu_encode(enc, data, opts, bs)
    assert again == full


def test_carry_fallback_identical_to_oracle(enc, monkeypatch):
    """What cannot be carried falls back to a state reset for the rest of the Block: a probability whose bounds
    have not met within the logged bits of a span.  With 1023 bits per span and slot that needs a strictly periodic bit sequence;
    reached on ordinary data by turning the cap down on both sides (XZAMD_TEST_LOG_CAP / orc_set_log_cap): the carry decisions,
    the bytes (identical to the oracle) and the round trip through liblzma."""
    import xz_amd
    data = xz_amd.corpus_text(5 << 20, seed=9).tobytes() + o.corpus_lorem(3 << 20)
    bs = 4 << 20
    opts = xz_amd.preset_options(6)
    opts.enc_span_bits = 400000
    prm = o.params_for_gpu_options(opts)
    full, _ = gpu_encode(enc, data, opts, bs)
    esb = bs // (256 << 10) + 1
    assert (enc.debug_fetch(16, 2 * esb).reshape(2, esb)[:, 1:4] == 1).all()          # (carried, with the real cap)
    try:
        for cap in (8, 300):
            monkeypatch.setenv("XZAMD_TEST_LOG_CAP", str(cap))
            o.orc_set_log_cap(cap)
            got, _ = gpu_encode(enc, data, opts, bs)
            gcarry = enc.debug_fetch(16, 2 * esb).reshape(2, esb)
            for b in range(2):
                blk = data[b * bs:(b + 1) * bs]
                starts, estarts = o.orc_piece_plan(blk, prm)
                ocarry = o.orc_two_phase_debug(blk, prm, len(starts), len(estarts))[3]
                assert (gcarry[b, 1:len(estarts)] == ocarry[1:]).all(), (cap, b, gcarry[b, :len(estarts)], ocarry)
                if cap == 8:
                    assert (ocarry[2:] == 0).all()                    # nothing merges within 8 bits: from the second span on, resets
            assert o.first_diff(got, o.orc_xz_stream(data, prm, bs)) == -1, cap
            rr, rdec = o.ref_decode(got, len(data) + 16)
            assert rr == 1 and rdec == data
            assert len(got) >= len(full)
    finally:
        o.orc_set_log_cap(0)
        monkeypatch.delenv("XZAMD_TEST_LOG_CAP", raising=False)
    again, _ = gpu_encode(enc, data, opts, bs)
    assert again == full


def test_output_independent_of_batching_and_feeding(enc):
    """The compressed bytes of presets 4-9 are a function of (input, options, block size) only: the same Stream whatever
    the device batch size (one batch, many small batches, the pipelined two-stream path) and however the client feeds
    lzma_code (one FINISH call, small LZMA_RUN chunks) -- like the reference, whose MT output does not depend on the
    thread count (round-3 advisor finding: the span plan used to depend on the batch)."""
    import xz_amd
    data = xz_amd.corpus_text(9 << 20, seed=21).tobytes() + o.corpus_lorem(2 << 20) + xz_amd.corpus_tar(3 << 20).tobytes()
    opts = xz_amd.preset_options(6)
    bs = 1 << 20
    ref, _ = gpu_encode(enc, data, opts, bs)
    assert enc.stats().batches == 1
    import torch
    for batch_mib in (2, 5):
        e2 = xz_amd.Encoder(0)
        try:
            e2.set_batch_bytes(batch_mib << 20)
            out, _ = gpu_encode(e2, data, opts, bs)
            assert e2.stats().batches > 2
            assert out == ref, ("batch size", batch_mib, o.first_diff(out, ref))
        finally:
            e2.close()
    # the lzma_* front end: one shot and in 64 KiB LZMA_RUN pieces
    import ctypes as C
    lib = xz_amd.lib()
    for piece in (len(data), 65536 + 13):
        out = _lzma_code_encode(lib, data, 6, bs, piece)
        assert out == ref, ("lzma_code feeding", piece, o.first_diff(out, ref))


def test_full_size_blocks_identical_to_oracle(enc):
    """Byte parity at the sizes the BASELINE configs use (the small parity cases never leave the first MiB of a Block):
    preset 6 on one full 24 MiB Block of the bench text (Block longer than its 8 MiB dictionary: the eligibility rule
    `distance <= dict_size` of the finder, cf. lz_encoder_mf.c:470-474, and ~190 parse pieces / ~40 encode spans), and
    preset 9e on a 40 MiB Block whose second half repeats the first with perturbations (64 MiB dictionary: matches at
    distances >= 2^23, i.e. the list format with the u16 length side array, 360-node windows, 256-byte suffix order).
    Every recorded symbol and every byte of the Stream vs the oracle, which runs in threads beside the GPU work."""
    import concurrent.futures as cf
    import xz_amd
    text = xz_amd.corpus_text(24 << 20, seed=1000)
    half = xz_amd.corpus_text(20 << 20, seed=77)
    rep = half.copy()
    rep[::4099] ^= 0x20                                    # the copy differs every 4 KiB: long matches at 20 MiB distance
    rep[5 << 20:6 << 20] = xz_amd.corpus_text(1 << 20, seed=78)
    big = np.concatenate([half, rep])
    cases = {"p6_text_24MiB": (text.tobytes(), 6), "p9e_repeat_40MiB": (big.tobytes(), 9 | 0x80000000)}
    with cf.ThreadPoolExecutor(max_workers=2) as pool:
        want = {}
        for name, (data, preset) in cases.items():
            opts = xz_amd.preset_options(preset)
            want[name] = pool.submit(o.orc_encode_block_syms, data, o.params_for_gpu_options(opts))
        for name, (data, preset) in cases.items():
            opts = xz_amd.preset_options(preset)
            bs = xz_amd.mt_block_size(opts)
            assert len(data) <= bs
            got, _ = gpu_encode(enc, data, opts, bs)
            n = len(data)
            gsl = enc.debug_fetch(9, n, "uint16")
            gsd = enc.debug_fetch(10, n, "uint32")
            payload, sl, sd = want[name].result()
            if preset & 0x80000000:
                far = (sl >= 2) & (sd >= (1 << 23))               # (the oracle's arrays are zero off the symbol starts)
                assert far.sum() > 1000, "the case must exercise distances >= 2^23"
            bad = _walk_symbols(sl, sd, gsl, gsd, n)
            assert bad is None, (name, "symbol records differ at", bad)
            prm = o.params_for_gpu_options(opts)
            ref_stream = o.orc_xz_stream(data, prm, bs, payloads=[payload])
            assert o.first_diff(got, ref_stream) == -1, (name, len(got), len(ref_stream))
            rr, rdec = o.ref_decode(got, n + 16)
            assert rr == 1 and rdec == data, name


SIZE_TOLERANCE = 0.025      # presets 4-9, full Blocks of 28 named classes: measured max +2.35 % at preset 6 (a cycled source tree; RGBA pixels
                            # +2.25 %, every other class <= +1.84 %) and +2.30 % at 9e (RGBA pixels) (round 6; round 5 stated 2.0 % over the 23
                            # classes it had been tuned on and measured +10.9 % on RGBA pixels, +3.8 % on varint records outside them)
SIZE_TOLERANCE_FAST = 0.01  # presets 1-3, default spans (256 KiB state-reset spans): measured max +0.72 % (HTML rows)


def _tolerance_cases(preset):
    """Full Blocks of every corpus class: the bench text, the reference's own text generator continued, x86-64 shared
    objects, a ustar stream of the image's source trees (BASELINE config C4's input) and of /opt/rocm/include alone
    (highly compressible headers: where state resets and a shallow suffix order cost the most)."""
    import xz_amd
    from test_oracle_encoder import _elf_mix

    def tar(n, roots=None):
        """None when the image has none of the source trees (the class is skipped, not failed)."""
        try:
            return (xz_amd.corpus_tar(n, roots) if roots else xz_amd.corpus_tar(n)).tobytes()
        except RuntimeError:
            return None
    if preset & 0x80000000:
        # 9e: 192 MiB Blocks, 64 MiB dictionary: one Block each; the tar stream is longer than the dictionary
        n_text, n_tar, n_elf = 16 << 20, 72 << 20, 16 << 20
        cases = {"bench_text": xz_amd.corpus_text(n_text, seed=1000).tobytes(), "tar": tar(n_tar)}
    else:
        n = 24 << 20                                # one full Block at presets 5 / 6
        n_elf = n
        cases = {"bench_text": xz_amd.corpus_text(n, seed=1000).tobytes(), "lorem": o.corpus_lorem(n),
                 "tar": tar(n), "rocm_headers": tar(n, "/opt/rocm/include")}
    cases = {k: v for k, v in cases.items() if v is not None}
    elf = _elf_mix(n_elf)
    if len(elf) == n_elf:
        cases["elf"] = elf
    # the classes the round-3 review measured outside the tolerance of that round (all of it span state resets): line-
    # structured logs, JSON records, a SQLite file, a tar stream of many small heterogeneous files
    import _corpora
    n_new = 16 << 20 if preset & 0x80000000 else 24 << 20
    cases["logs"] = _corpora.logs(n_new)
    cases["json"] = _corpora.json_records(n_new)
    cases["sqlite"] = _corpora.sqlite_file(n_new)
    dpkg = _corpora.dpkg_tar(n_new)
    if dpkg is not None:
        cases["dpkg_tar"] = dpkg
    # round 5: the literal-heavy / numeric classes the round-4 review measured outside the tolerance (float32 arrays +4 ...
    # +5.4 %: the seeded price model at piece starts), all seeded numpy / random generators -- no dependence on the image
    for name, gen in _corpora.NUMERIC_CLASSES.items():
        cases[name] = gen(n_new)
    meta = _corpora.elf_metadata(n_new)      # round 5: +4.3 ... +4.9 % before the coder kept the parser's rep / match choice
    if meta is not None:
        cases["elf_metadata"] = meta
    # round 6: the classes the round-5 review probed and found outside (image pixels +10.9 %, varint records +3.8 %, a cycled
    # source tree +2.2 %, CJK text +1.9 %: every piece settled into a worse regime than a continuous parse)
    for name, gen in _corpora.REVIEW_CLASSES.items():
        d = gen(n_new)
        if d is not None:
            cases[name] = d
    for name, (gen, _) in _corpora.KNOWN_OUTSIDE.items():          # measured and pinned with their own (looser) bound
        cases[name] = gen(n_new)
    return cases


@pytest.mark.parametrize("preset", [6, 9 | 0x80000000])
def test_size_within_tolerance_of_reference(enc, preset):
    """Stated tolerance (README/DESIGN): at the same preset and Block size the device output is at most SIZE_TOLERANCE
    larger than the REAL liblzma's (oracle/_ref), on FULL Blocks (24 MiB at -6) of every corpus class."""
    import concurrent.futures as cf
    import xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    cases = _tolerance_cases(preset)
    opts = xz_amd.preset_options(preset)
    bs = xz_amd.mt_block_size(opts)
    report = {}
    with cf.ThreadPoolExecutor(max_workers=len(cases)) as pool:      # the reference encodes run beside the GPU work
        refs = {name: pool.submit(o.ref_encode_mt, data, preset, 2, bs) for name, data in cases.items()}
        outs = {name: gpu_encode(enc, data, opts, bs)[0] for name, data in cases.items()}
        for name, data in cases.items():
            got, ref = outs[name], refs[name].result()
            r, dec = o.ref_decode(got, len(data) + 16)
            assert r == 1 and dec == data, ("liblzma decoder", name)
            report[name] = round(100.0 * (len(got) / len(ref) - 1), 2)
    print("size vs liblzma, preset", hex(preset), report)
    import _corpora
    over = {k: v for k, v in report.items() if v > 100.0 * (_corpora.KNOWN_OUTSIDE[k][1] if k in _corpora.KNOWN_OUTSIDE else SIZE_TOLERANCE)}
    assert not over, (hex(preset), report)


RANDOM_TABLE_DRAWS = 32


def test_size_distribution_over_random_record_tables(enc):
    """The stated tolerance as a PROPERTY, not a list (round-5 review): full 24 MiB Blocks of RANDOM_TABLE_DRAWS draws of the
    seeded random class generator (tests/_corpora.random_class: tables of records of width 1 .. 64 with counter / random-walk /
    enum / noise-bit / text / constant fields, a third of them mixtures of two tables) through the product path at preset 6,
    against the real liblzma on the same Block.  Tables of records are where two codings of the same bytes cost about the same
    under their own adapted model, and which one an encoder settles into decides +-10 %: the distribution is what is pinned --
    its median, the share of the draws inside the stated tolerance -- and printed in full."""
    import concurrent.futures as cf
    import xz_amd
    import _corpora
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    opts = xz_amd.preset_options(6)
    bs = xz_amd.mt_block_size(opts)
    n = 24 << 20
    report = {}
    with cf.ThreadPoolExecutor(max_workers=16) as pool:
        datas = {s: _corpora.random_class(s, n) for s in range(RANDOM_TABLE_DRAWS)}
        refs = {s: pool.submit(o.ref_encode_mt, d, 6, 2, bs) for s, d in datas.items()}
        for s, d in datas.items():
            got = gpu_encode(enc, d, opts, bs)[0]
            r, dec = o.ref_decode(got, n + 16)
            assert r == 1 and dec == d, ("liblzma decoder", s)
            report[s] = round(100.0 * (len(got) / len(refs[s].result()) - 1), 2)
    v = np.array(list(report.values()))
    inside = float((v <= 100.0 * SIZE_TOLERANCE).mean())
    print("size vs liblzma over random record tables, preset 6:", report, "median", float(np.median(v)), "inside", inside,
          "p90", float(np.percentile(v, 90)), "max", float(v.max()), "min", float(v.min()))
    assert float(np.median(v)) <= 1.0, report
    assert inside >= 0.70, report


@pytest.mark.parametrize("preset", [1, 3])
def test_size_within_tolerance_of_reference_fast_presets(enc, preset):
    """Presets 1-3 in PRODUCT mode (default spans: 256 KiB state-reset spans; one span per Block is byte-identical to
    liblzma and pinned elsewhere): output within 1 % of liblzma's on full Blocks of the text-like classes where a reset
    costs the most (round-4 review: HTML rows +2.82 % with 64 KiB spans) and on binary / numeric data."""
    import concurrent.futures as cf
    import xz_amd
    import _corpora
    from test_oracle_encoder import _elf_mix
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    opts = xz_amd.preset_options(preset)
    bs = xz_amd.mt_block_size(opts)
    n = 2 * bs
    cases = {"bench_text": xz_amd.corpus_text(n, seed=1000).tobytes(), "lorem": o.corpus_lorem(n), "html": _corpora.html_rows(n),
             "logs": _corpora.logs(n), "f32sine": _corpora.f32_sine(n), "csv": _corpora.csv_sensors(n)}
    elf = _elf_mix(n)
    if len(elf) == n:
        cases["elf"] = elf
    report = {}
    with cf.ThreadPoolExecutor(max_workers=len(cases)) as pool:
        refs = {name: pool.submit(o.ref_encode_mt, data, preset, 2, bs) for name, data in cases.items()}
        for name, data in cases.items():
            got, _ = gpu_encode(enc, data, opts, bs)
            rr, dec = o.ref_decode(got, len(data) + 16)
            assert rr == 1 and dec == data, ("liblzma decoder", name)
            report[name] = round(100.0 * (len(got) / len(refs[name].result()) - 1), 2)
    print("size vs liblzma, preset", preset, report)
    over = {k: v for k, v in report.items() if v > 100.0 * SIZE_TOLERANCE_FAST}
    assert not over, (preset, report)


@pytest.mark.parametrize("preset", [0, 1, 3])
def test_x86_bcj_chain_identical_to_reference(enc, preset):
    """Chain {x86 BCJ, LZMA2} (simple/x86.c): with one span per Block the whole .xz Stream equals the
    reference MT encoder's, Block Headers with two filters included; several Blocks, a Block whose
    size is not a multiple of the BCJ chunk, and input with long stretches without synchronisation
    points."""
    import xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    opts = xz_amd.preset_options(preset, span_size=xz_amd.SPAN_WHOLE_BLOCK)
    opts.bcj = xz_amd.BCJ_X86
    cases = {"x86": o.corpus_x86(700000, 4), "dense": o.corpus_x86(300000, 5, density=6),
             "text": o.corpus_lorem(100000), "tiny4": bytes([0xE8, 0, 0, 0]), "tiny5": bytes([0xE8, 1, 0, 0, 0]),
             "all_e8": bytes([0xE8]) * 70000, "rnd": o.corpus_random()}
    for name, data in cases.items():
        for bs in (1 << 20, 200000, 65537):
            got, _ = gpu_encode(enc, data, opts, bs)
            ref = o.ref_encode_mt_x86(data, preset, threads=2, block_size=bs)
            assert o.first_diff(got, ref) == -1, (name, preset, bs)


def _arm64_like(n, seed):
    """Bytes with plenty of BL (0x94......) and ADRP (0x90/0xB0/0xD0/0xF0 ......) shaped words at aligned offsets."""
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 1 << 32, n // 4 + 1, dtype=np.uint64).astype(np.uint32)
    sel = rng.random(len(w))
    w[sel < 0.3] = (w[sel < 0.3] & 0x03FFFFFF) | 0x94000000
    adrp = (sel >= 0.3) & (sel < 0.5)
    w[adrp] = (w[adrp] & 0x60FFFFFF) | 0x90000000
    w[adrp & (rng.random(len(w)) < 0.7)] &= 0xFF03FFFF       # small immediates: inside the +/-512 MiB range
    return w.astype("<u4").tobytes()[:n]


_SIMPLE_BCJ = {"powerpc": 5, "ia64": 6, "arm": 7, "armthumb": 8, "sparc": 9, "riscv": 0x0B}


def _bcj_like(kind, n, seed):
    """Random bytes with many branch-shaped words of the given architecture planted at aligned offsets."""
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, n + 16, dtype=np.uint8)
    if kind == "arm":
        idx = np.arange(0, n - 4, 4)[rng.random((n - 4 + 3) // 4) < 0.3]
        b[idx + 3] = 0xEB
    elif kind == "powerpc":
        idx = np.arange(0, n - 4, 4)[rng.random((n - 4 + 3) // 4) < 0.3]
        b[idx] = 0x48 | (b[idx] & 3)
        b[idx + 3] = (b[idx + 3] & 0xFC) | 1
    elif kind == "sparc":
        idx = np.arange(0, n - 4, 4)[rng.random((n - 4 + 3) // 4) < 0.3]
        pos = rng.random(len(idx)) < 0.5
        b[idx[pos]] = 0x40
        b[idx[pos] + 1] &= 0x3F
        b[idx[~pos]] = 0x7F
        b[idx[~pos] + 1] |= 0xC0
    elif kind == "armthumb":
        idx = np.arange(0, n - 4, 2)[rng.random((n - 4 + 1) // 2) < 0.2]
        b[idx + 1] = 0xF0 | (b[idx + 1] & 7)
        b[idx + 3] = 0xF8 | (b[idx + 3] & 7)
    elif kind == "ia64":
        for i in range(0, n - 16, 16):
            if rng.random() < 0.5:
                v = int.from_bytes(b[i:i + 16].tobytes(), "little")
                v = (v & ~0x1F) | int(rng.choice([16, 17, 18, 19, 22, 23, 24, 25, 28, 29]))
                for slot in range(3):
                    if rng.random() < 0.7:
                        bp = 5 + 41 * slot
                        v &= ~(0xF << (bp + 37)); v |= 5 << (bp + 37)
                        v &= ~(7 << (bp + 9))
                b[i:i + 16] = np.frombuffer((v & ((1 << 128) - 1)).to_bytes(16, "little"), dtype=np.uint8)
    elif kind == "riscv":
        # JAL (rd x1/x5 and others), AUIPC pairs with matching and non-matching second instructions, AUIPC with
        # rd x0/x2 in and out of the encoder's special form: every jump length of the reference's walk (2/4/6/8)
        for i in range(0, n - 8, 2):
            r = rng.random()
            if r < 0.08:
                b[i] = 0xEF
                if rng.random() < 0.7:
                    b[i + 1] &= 0xF2
            elif r < 0.2:
                inst = (int(rng.integers(0, 1 << 32)) & ~0x7F) | 0x17
                if rng.random() < 0.5:
                    rd = int(rng.choice([1, 3, 5, 6, 10, 31]))
                    inst = (inst & ~(0x1F << 7)) | (rd << 7)
                    if rng.random() < 0.7:
                        i2 = (int(rng.integers(0, 1 << 32)) & ~(0x1F << 15)) | (rd << 15) | 3
                        b[i + 4:i + 8] = np.frombuffer(i2.to_bytes(4, "little"), dtype=np.uint8)
                else:
                    inst = (inst & ~0x3FFF) | 0x3117 if rng.random() < 0.7 else (inst & ~(0x1F << 7)) | (int(rng.choice([0, 2])) << 7)
                    if rng.random() < 0.3:
                        inst &= 0x07FFFFFF
                b[i:i + 4] = np.frombuffer((inst & 0xFFFFFFFF).to_bytes(4, "little"), dtype=np.uint8)
    return b[:n].tobytes()


@pytest.mark.parametrize("kind", sorted(_SIMPLE_BCJ))
def test_simple_bcj_chains_identical_to_reference(enc, kind):
    """{PowerPC | IA-64 | ARM | ARM-Thumb | SPARC | RISC-V BCJ, LZMA2} (simple/powerpc.c, ia64.c, arm.c, armthumb.c, sparc.c,
    riscv.c):
    with one span per Block the whole .xz Stream equals the reference MT encoder's, Blocks that are not multiples of
    the instruction size, tiny inputs; in span mode it decodes bit-exactly through the real decoder."""
    import torch, xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    fid = _SIMPLE_BCJ[kind]
    cases = {"code": _bcj_like(kind, 120000 if kind == "riscv" else 500000, 5), "mixed": o.corpus_mixed(200000, 4), "tiny": _bcj_like(kind, 3, 1),
             "odd": _bcj_like(kind, 4099, 2), "17": _bcj_like(kind, 17, 3)}
    for preset in (1, 3):
        opts = xz_amd.preset_options(preset, span_size=xz_amd.SPAN_WHOLE_BLOCK)
        opts.bcj = fid
        for name, data in cases.items():
            for bs in (1 << 20, 200001, 65537):
                t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                out, _ = enc.encode(t, opts=opts, block_size=bs, check=4)
                ref = o.ref_encode_mt_chain(data, preset, fid, 1, threads=2, block_size=bs, check=4)
                assert o.first_diff(out.cpu().numpy().tobytes(), ref) == -1, (kind, name, preset, bs)
    opts = xz_amd.preset_options(6)
    opts.bcj = fid
    data = cases["code"] + cases["mixed"]
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out, _ = enc.encode(t, opts=opts, block_size=300000, check=4)
    rr, dec = o.ref_decode(out.cpu().numpy().tobytes(), len(data) + 16)
    assert rr == 1 and dec == data, kind


@pytest.mark.parametrize("kind", ["arm64", "delta1", "delta4", "delta256", "sha256"])
def test_arm64_delta_chains_and_sha256_identical_to_reference(enc, kind):
    """{ARM64 BCJ, LZMA2} (simple/arm64.c), {delta, LZMA2} (delta/delta_encoder.c) and the SHA-256 Check
    (check/sha256.c): with one span per Block the whole .xz Stream equals the reference MT encoder's; in span mode
    it decodes bit-exactly through the real decoder.  Blocks that are not multiples of 4, tiny inputs."""
    import xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    cases = {"arm": _arm64_like(600000, 3), "mixed": o.corpus_mixed(300000, 9), "tiny": bytes([0x94, 1, 2]), "five": b"\x00\x00\x00\x94\x07"}
    for preset in (1, 3):
        opts = xz_amd.preset_options(preset, span_size=xz_amd.SPAN_WHOLE_BLOCK)
        check = 4
        fid, dist = 0, 1
        if kind == "arm64":
            opts.bcj, fid = xz_amd.BCJ_ARM64, 0x0A
        elif kind.startswith("delta"):
            dist = int(kind[5:])
            opts.bcj, fid = xz_amd.filter_delta(dist), 0x03
        else:
            check = 10
        for name, data in cases.items():
            for bs in (1 << 20, 200001, 65537):
                t = __import__("torch").frombuffer(bytearray(data), dtype=__import__("torch").uint8).cuda()
                out, _ = enc.encode(t, opts=opts, block_size=bs, check=check)
                got = out.cpu().numpy().tobytes()
                if fid:
                    ref = o.ref_encode_mt_chain(data, preset, fid, dist, threads=2, block_size=bs, check=check)
                else:
                    ref = o.ref_encode_mt(data, preset, threads=2, block_size=bs, check=check)
                assert o.first_diff(got, ref) == -1, (kind, name, preset, bs)
    opts = xz_amd.preset_options(6)
    if kind == "arm64":
        opts.bcj = xz_amd.BCJ_ARM64
    elif kind.startswith("delta"):
        opts.bcj = xz_amd.filter_delta(int(kind[5:]))
    data = cases["arm"] + cases["mixed"]
    t = __import__("torch").frombuffer(bytearray(data), dtype=__import__("torch").uint8).cuda()
    out, _ = enc.encode(t, opts=opts, block_size=300000, check=10 if kind == "sha256" else 4)
    rr, dec = o.ref_decode(out.cpu().numpy().tobytes(), len(data) + 16)
    assert rr == 1 and dec == data, kind


_CHAINS = {
    "delta4_x86": [(0x03, 4), (0x04, 0)],
    "x86_delta1": [(0x04, 0), (0x03, 1)],
    "delta2_x86_arm64": [(0x03, 2), (0x04, 0), (0x0A, 0)],
    "arm64_arm64_arm64": [(0x0A, 0), (0x0A, 0), (0x0A, 0)],
    "riscv_delta256_powerpc": [(0x0B, 0), (0x03, 256), (0x05, 0)],
}


@pytest.mark.parametrize("name", sorted(_CHAINS))
def test_filter_chains_of_up_to_four_identical_to_reference(enc, name, monkeypatch):
    """Chains of two and three filters in front of LZMA2 (common/filter_common.c:250-334: at most four filters, LZMA2
    last; each BCJ / delta filter runs over what the one before it made, simple_coder.c / delta_encoder.c): with one span
    per Block the whole .xz Stream (Block Headers with every Filter Flags field incl.) equals the reference MT encoder's,
    through xzamd_stream_encode and through lzma_stream_encoder_mt with the same lzma_filter array; the default
    two-phase encode decodes bit-exactly through the real decoder."""
    import torch, xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    chain = _CHAINS[name]

    def put(opts):
        vals = [xz_amd.filter_delta(d) if fid == 3 else fid for fid, d in chain] + [0, 0]
        opts.bcj, opts.bcj2, opts.bcj3 = vals[0], vals[1], vals[2]
        return opts
    cases = {"x86": o.corpus_x86(400000, 3), "arm": _arm64_like(300000, 3), "mixed": o.corpus_mixed(200000, 4), "tiny": bytes([0x94, 1, 2]), "odd": _arm64_like(4099, 2)}
    for preset in (1, 3):
        opts = put(xz_amd.preset_options(preset, span_size=xz_amd.SPAN_WHOLE_BLOCK))
        for cname, data in cases.items():
            for bs in (1 << 20, 200001, 65537):
                t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                out, _ = enc.encode(t, opts=opts, block_size=bs, check=4)
                ref = o.ref_encode_mt_chain_n(data, preset, chain, threads=2, block_size=bs, check=4)
                assert o.first_diff(out.cpu().numpy().tobytes(), ref) == -1, (name, cname, preset, bs)
    data = cases["x86"] + cases["arm"] + cases["mixed"]
    # the same chain as a lzma_filter array through the liblzma entry points (preset 1's lzma_options_lzma)
    import ctypes as C
    import sys
    if os.path.join(o.ROOT, "tools") not in sys.path:
        sys.path.insert(0, os.path.join(o.ROOT, "tools"))
    from bench_lzma_code import Mt, Stream

    class Filter(C.Structure):
        _fields_ = [("id", C.c_uint64), ("options", C.c_void_p)]

    class OptLzma(C.Structure):
        _fields_ = [("dict_size", C.c_uint32), ("preset_dict", C.c_void_p), ("preset_dict_size", C.c_uint32),
                    ("lc", C.c_uint32), ("lp", C.c_uint32), ("pb", C.c_uint32), ("mode", C.c_int),
                    ("nice_len", C.c_uint32), ("mf", C.c_int), ("depth", C.c_uint32), ("pad", C.c_uint8 * 64)]
    ol = OptLzma(dict_size=1 << 20, lc=3, lp=0, pb=2, mode=1, nice_len=128, mf=4, depth=8)
    deltas = [(C.c_uint32 * 8)(0, d) for _, d in chain]          # lzma_options_delta: type BYTE, dist
    fl = (Filter * (len(chain) + 2))()
    for i, (fid, _) in enumerate(chain):
        fl[i].id, fl[i].options = fid, (C.cast(deltas[i], C.c_void_p) if fid == 3 else None)
    fl[len(chain)].id, fl[len(chain)].options = 0x21, C.cast(C.pointer(ol), C.c_void_p)
    fl[len(chain) + 1].id = 2**64 - 1
    L = xz_amd.lib()
    ib = C.create_string_buffer(data, len(data))
    ob = C.create_string_buffer(len(data) + (1 << 20))
    st = Stream()
    m = Mt(threads=1, check=4, block_size=200001, filters=C.cast(fl, C.c_void_p))
    monkeypatch.setenv("XZAMD_SPAN_KIB", "1024")           # one span per Block: the reference's bytes
    assert L.lzma_stream_encoder_mt(C.byref(st), C.byref(m)) == 0
    monkeypatch.delenv("XZAMD_SPAN_KIB")
    st.next_in, st.avail_in = C.cast(ib, C.c_void_p).value, len(data)
    st.next_out, st.avail_out = C.cast(ob, C.c_void_p).value, len(ob)
    r = L.lzma_code(C.byref(st), 3)
    while r == 0:
        r = L.lzma_code(C.byref(st), 3)
    assert r == 1
    got = ob.raw[: st.total_out]
    L.lzma_end(C.byref(st))
    ref = o.ref_encode_mt_chain_n(data, 1, chain, threads=2, block_size=200001, check=4)
    assert o.first_diff(got, ref) == -1, (name, "lzma_code")

    opts = put(xz_amd.preset_options(6))
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    out, _ = enc.encode(t, opts=opts, block_size=300000, check=4)
    rr, dec = o.ref_decode(out.cpu().numpy().tobytes(), len(data) + 16)
    assert rr == 1 and dec == data, name


@pytest.mark.parametrize("preset", [1, 6, 9 | 0x80000000])
def test_x86_bcj_default_spans_roundtrip(enc, preset):
    """Span-parallel mode with the BCJ pre-pass: decodes bit-exactly through the REAL reference decoder,
    and the LZMA2 payload is what the oracle makes of the reference-filtered Block."""
    import xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    opts = xz_amd.preset_options(preset)
    opts.bcj = xz_amd.BCJ_X86
    data = o.corpus_x86(900000, 9)
    bs = 400000
    got, binfo = gpu_encode(enc, data, opts, bs)
    rr, dec = o.ref_decode(got, len(data) + 16)
    assert rr == 1 and dec == data
    assert len(binfo) == 3
    # plain LZMA2 of the same data must differ (the filter really ran) ...
    plain = xz_amd.preset_options(preset)
    got_plain, _ = gpu_encode(enc, data, plain, bs)
    assert got_plain != got
    # ... and encoding the reference-filtered Blocks with the plain chain gives the same LZMA2 payload
    filt = b"".join(o.ref_x86_filter(data[i:i + bs]) for i in range(0, len(data), bs))
    got_f, binfo_f = gpu_encode(enc, filt, plain, bs)
    assert len(binfo_f) == len(binfo)
    for a, b in zip(binfo, binfo_f):
        blk_a = got[a.out_offset:a.out_offset + a.total_size]
        blk_b = got_f[b.out_offset:b.out_offset + b.total_size]
        hs_a, hs_b = (blk_a[0] + 1) * 4, (blk_b[0] + 1) * 4
        assert hs_a >= hs_b and blk_a[1] == 0xC1 and blk_b[1] == 0xC0
        assert blk_a[hs_a:-8] == blk_b[hs_b:-8]          # LZMA2 payload + padding; the Checks differ by design


@pytest.mark.parametrize("check", [0, 1, 4])
def test_block_checks_identical_to_reference(enc, check):
    """None / CRC32 / CRC64 Block checks (check/crc32_fast.c, crc64_fast.c): whole Stream equals the
    reference MT encoder's; Blocks longer than one CRC strip, a short last Block, an empty-ish one."""
    import torch
    import xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    opts = xz_amd.preset_options(1, span_size=xz_amd.SPAN_WHOLE_BLOCK)
    for data in (o.corpus_mixed(700000, 8), o.corpus_lorem(4097), b"z"):
        for bs in (1 << 20, 100000):
            t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            out, _ = enc.encode(t, opts=opts, block_size=bs, check=check)
            got = out.cpu().numpy().tobytes()
            ref = o.ref_encode_mt(data, 1, threads=2, block_size=bs, check=check)
            assert o.first_diff(got, ref) == -1, (check, len(data), bs)


def test_custom_options_and_small_dictionary(enc):
    import xz_amd
    data = o.corpus_mixed(500000, 8)
    for dict_size, lc, lp, pb, nice, mf, depth in [
            (4096, 3, 0, 2, 32, 4, 4), (65536, 0, 2, 0, 273, 4, 1), (1 << 20, 3, 0, 4, 8, 3, 6),
            (12345, 1, 2, 1, 64, 4, 56), (1 << 16, 3, 0, 2, 32, 3, 12)]:
        opts = xz_amd.LzmaOptions(dict_size, lc, lp, pb, 1, nice, mf, depth, mf, nice, depth, xz_amd.SPAN_WHOLE_BLOCK)
        got, _ = gpu_encode(enc, data, opts, 1 << 20)
        want = o.orc_xz_stream(data, o.params_for_gpu_options(opts), 1 << 20)
        assert o.first_diff(got, want) == -1, (dict_size, lc, lp, pb, nice, mf, depth)


def test_empty_input(enc):
    import xz_amd
    opts = xz_amd.preset_options(6)
    got, binfo = gpu_encode(enc, b"", opts, 0)
    r, dec, nb = o.orc_xz_decode(got, 16)
    assert r == 0 and dec == b"" and nb == 0 and binfo == []
    if o.have_ref():
        assert got == o.ref_encode_mt(b"", 6, threads=2)


def test_block_infos_and_blocks_only(enc):
    import torch
    import xz_amd
    data = o.corpus_lorem(700000)
    opts = xz_amd.preset_options(1)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    whole, bi = enc.encode(t, opts=opts, block_size=1 << 18)
    whole = whole.cpu().numpy().tobytes()
    blocks, bi2 = enc.encode(t, opts=opts, block_size=1 << 18, blocks_only=True)
    blocks = blocks.cpu().numpy().tobytes()
    assert len(bi) == len(bi2) == 3
    assert whole[12:12 + len(blocks)] == blocks
    assert [b.total_size for b in bi] == [b.total_size for b in bi2]
    assert sum(b.uncompressed_size for b in bi) == len(data)


def test_large_roundtrip_properties(enc):
    """Full-size property check (no oracle encode: it would take minutes): 1 GiB of synthetic text,
    preset 6 mapping, default spans -> decoded sha256 must equal the input's; sizes self-consistent."""
    import torch
    import xz_amd
    n = 1 << 30
    host = xz_amd.corpus_text(n, seed=7)
    t = torch.from_numpy(host).cuda()
    opts = xz_amd.preset_options(6)
    out, binfo = enc.encode(t, opts=opts)
    got = out.cpu().numpy().tobytes()
    assert len(binfo) == (n + (24 << 20) - 1) // (24 << 20)
    assert sum(b.uncompressed_size for b in binfo) == n
    # sample Blocks are decoded by the oracle; the whole stream by the real reference (fast C)
    if o.have_ref():
        r, dec = o.ref_decode(got, n + 16)
        assert r == 1
        assert hashlib.sha256(dec).digest() == hashlib.sha256(host.tobytes()).digest()
    else:
        r, dec, nb = o.orc_xz_decode(got, n + 16)
        assert r == 0 and nb == len(binfo)
        assert hashlib.sha256(dec).digest() == hashlib.sha256(host.tobytes()).digest()


def test_batches_and_chain_build_overlap_do_not_change_the_stream(monkeypatch):
    """Several device batches: the next batch's chain build runs on a low-priority stream underneath the
    span kernel of the current one.  Output must equal the single-batch Stream, with and without overlap."""
    import torch
    import xz_amd
    data = xz_amd.corpus_text(48 << 20, seed=9)
    t = torch.from_numpy(data).cuda()
    for preset in (6, 1):
        opts = xz_amd.preset_options(preset)
        e = xz_amd.Encoder(0)
        want, _ = e.encode(t, opts=opts, block_size=1 << 20)
        want = want.cpu().numpy().tobytes()
        assert e.stats().batches == 1
        e.set_batch_bytes(8 << 20)
        got, _ = e.encode(t, opts=opts, block_size=1 << 20)
        assert e.stats().batches == 6
        assert got.cpu().numpy().tobytes() == want, preset
        monkeypatch.setenv("XZAMD_NO_OVERLAP", "1")
        got2, _ = e.encode(t, opts=opts, block_size=1 << 20)
        assert got2.cpu().numpy().tobytes() == want, preset
        monkeypatch.delenv("XZAMD_NO_OVERLAP")
        e.close()


def test_out_of_memory_falls_back_to_smaller_batches(monkeypatch):
    """A failed device allocation halves the batch (whole Blocks) instead of failing the encode; the Stream
    does not depend on how Blocks were batched."""
    import torch
    import xz_amd
    data = xz_amd.corpus_text(24 << 20, seed=3)
    t = torch.from_numpy(data).cuda()
    opts = xz_amd.preset_options(6)
    e1 = xz_amd.Encoder(0)
    want, _ = e1.encode(t, opts=opts, block_size=1 << 20)
    want = want.cpu().numpy().tobytes()
    assert e1.stats().batches == 1
    e1.close()
    monkeypatch.setenv("XZAMD_TEST_ALLOC_LIMIT_MIB", "300")     # match lists need 64 B per input byte
    e2 = xz_amd.Encoder(0)
    got, _ = e2.encode(t, opts=opts, block_size=1 << 20)
    assert e2.stats().batches > 1
    assert got.cpu().numpy().tobytes() == want
    e2.close()
    monkeypatch.setenv("XZAMD_TEST_ALLOC_LIMIT_MIB", "1")       # not even one Block fits: a clean error
    e3 = xz_amd.Encoder(0)                                      # (the hook is read when a context is created)
    with pytest.raises(xz_amd.XzAmdError):
        e3.encode(t, opts=opts, block_size=1 << 20)
    e3.close()


def test_missing_gpu_fails_loudly():
    """The product path has no CPU fallback (checked structurally: ctx_create fails without a device)."""
    import ctypes as C
    import xz_amd
    ctx = C.c_void_p()
    assert xz_amd.lib().xzamd_ctx_create(C.byref(ctx), 9999) != 0


# ---------------------------------------------------------------------------------------------
# The liblzma drop-in boundary: lzma_stream_encoder_mt / lzma_code / lzma_end / lzma_get_progress
# ---------------------------------------------------------------------------------------------
def _build_client(tmp_path):
    import os
    import subprocess
    root = o.ROOT
    exe = os.path.join(str(tmp_path), "compress_mt")
    subprocess.run(["gcc", "-O2", "-DUSE_XZ_AMD", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "compress_mt.c"), "-o", exe,
                    "-L" + os.path.join(root, "xz_amd"), "-lxz_amd",
                    "-Wl,-rpath," + os.path.join(root, "xz_amd")], check=True)
    return exe


@pytest.mark.parametrize("preset,bs,iobuf", [(1, 65536, 8192), (6, 0, 1 << 20), (3, 200000, 777)])
def test_liblzma_client_through_lzma_code(tmp_path, preset, bs, iobuf):
    """A plain liblzma client (examples/compress_mt.c, the 04_compress_easy_mt.c pattern) linked
    against libxz_amd.so: output must decode bit-exactly with the reference decoder and with the
    system xz, and equal the batch API's stream."""
    import subprocess
    import xz_amd
    exe = _build_client(tmp_path)
    data = o.corpus_lorem(700001) + bytes(np.random.default_rng(3).integers(0, 256, size=50000, dtype=np.uint8))
    p = subprocess.run([exe, str(preset), str(bs), str(iobuf)], input=data, capture_output=True, check=True)
    got = p.stdout
    r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    assert r == 0 and dec == data
    opts = xz_amd.preset_options(preset)
    want = o.orc_xz_stream(data, o.params_for_gpu_options(opts), bs or xz_amd.mt_block_size(opts))
    assert o.first_diff(got, want) == -1
    sysxz = subprocess.run(["xz", "-dc"], input=got, capture_output=True)
    if sysxz.returncode == 0:                      # stock xz 5.2.5 of the image
        assert sysxz.stdout == data


@pytest.mark.parametrize("level", [1, 6])
def test_ld_preload_interposer_with_stock_xz(tmp_path, level):
    """The unmodified `xz` binary of the image (5.2.5, dynamically linked against its own liblzma) with
    LD_PRELOAD=libxz_amd_preload.so: its multi-threaded encoder runs on the GPU, everything else stays in
    the real liblzma.  Output must equal the device API's Stream for the same preset and decode with the
    plain binary."""
    import os
    import shutil
    import subprocess
    import torch
    import xz_amd
    xz = shutil.which("xz")
    pre = os.path.join(os.path.dirname(xz_amd.LIB_PATH), "libxz_amd_preload.so")
    if not xz:
        pytest.skip("no xz binary")
    assert os.path.exists(pre), "libxz_amd_preload.so not built"
    data = xz_amd.corpus_text(40 << 20, seed=5).tobytes()[: (33 << 20) + 12345]
    src = tmp_path / "input.bin"
    src.write_bytes(data)
    env = dict(os.environ, LD_PRELOAD=pre, XZ_AMD_VERBOSE="1")
    p = subprocess.run([xz, "-T4", f"-{level}", "-c", str(src)], capture_output=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"lzma_stream_encoder_mt -> GPU" in p.stderr, p.stderr.decode()[-2000:]
    got = p.stdout
    # plain xz (no preload) decodes it
    d = subprocess.run([xz, "-dc"], input=got, capture_output=True, timeout=600)
    assert d.returncode == 0 and d.stdout == data
    # and the preload does not disturb decoding either
    d2 = subprocess.run([xz, "-dc"], input=got, capture_output=True, env=env, timeout=600)
    assert d2.returncode == 0 and d2.stdout == data
    # same bytes as the device API with the same preset
    enc = xz_amd.Encoder(0)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    want, _ = enc.encode(t, preset=level)
    assert o.first_diff(got, want.cpu().numpy().tobytes()) == -1
    enc.close()
    # XZ_AMD_DISABLE routes to the real liblzma
    env2 = dict(env, XZ_AMD_DISABLE="1")
    p2 = subprocess.run([xz, "-T4", f"-{level}", "-c", str(src)], capture_output=True, env=env2, timeout=900)
    assert p2.returncode == 0 and b"-> GPU" not in p2.stderr


def test_ld_preload_block_size_and_block_list(tmp_path):
    """`xz --block-size` (lzma_mt.block_size) and `xz --block-list` (LZMA_FULL_FLUSH at the listed sizes,
    src/xz/coder.c:1163-1218) through the interposer: Block boundaries where the options put them, same bytes as the
    device API with that Block size, bit-exact through the plain binary."""
    import shutil
    import subprocess
    import torch
    import xz_amd
    xz = shutil.which("xz")
    pre = os.path.join(os.path.dirname(xz_amd.LIB_PATH), "libxz_amd_preload.so")
    if not xz:
        pytest.skip("no xz binary")
    data = xz_amd.corpus_text(8 << 20, seed=6).tobytes()[: (7 << 20) + 4321]
    src = tmp_path / "input.bin"
    src.write_bytes(data)
    env = dict(os.environ, LD_PRELOAD=pre, XZ_AMD_VERBOSE="1")

    def blocks_of(stream):
        r, dec, nb = o.orc_xz_decode(stream, len(data) + 16)
        assert r == 0 and dec == data
        return nb

    p = subprocess.run([xz, "-T4", "-1", "--block-size=2MiB", "-c", str(src)], capture_output=True, env=env, timeout=600)
    assert p.returncode == 0 and b"-> GPU" in p.stderr, p.stderr.decode()[-2000:]
    assert blocks_of(p.stdout) == 4
    enc = xz_amd.Encoder(0)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    want, _ = enc.encode(t, preset=1, block_size=2 << 20)
    enc.close()
    assert o.first_diff(p.stdout, want.cpu().numpy().tobytes()) == -1
    # --block-list: 1 MiB, 3 MiB, then the rest in Blocks of the default size (3 MiB at -1): 1 + 3 + 3 + 0.004 MiB
    p = subprocess.run([xz, "-T4", "-1", "--block-list=1MiB,3MiB,0", "-c", str(src)], capture_output=True, env=env, timeout=600)
    assert p.returncode == 0 and b"-> GPU" in p.stderr, p.stderr.decode()[-2000:]
    got = p.stdout
    assert blocks_of(got) == 4
    lst = subprocess.run([xz, "--robot", "--list", "-vv"], input=b"", capture_output=True)     # availability probe only
    f = tmp_path / "bl.xz"
    f.write_bytes(got)
    lst = subprocess.run([xz, "--robot", "--list", "-vv", str(f)], capture_output=True, text=True)
    if lst.returncode == 0:
        usizes = [int(l.split("\t")[7]) for l in lst.stdout.splitlines() if l.startswith("block\t")]
        assert usizes[:2] == [1 << 20, 3 << 20] and sum(usizes) == len(data), usizes
    d = subprocess.run([xz, "-dc", str(f)], capture_output=True, timeout=600)
    assert d.returncode == 0 and d.stdout == data


def test_one_shot_buffer_api():
    """lzma_easy_buffer_encode / lzma_stream_buffer_encode / lzma_stream_buffer_bound
    (common/easy_buffer_encoder.c, stream_buffer_encoder.c): same bytes as the streaming API, BUF_ERROR
    without touching *out_pos when the output does not fit."""
    import ctypes as C
    import xz_amd
    L = xz_amd.lib()
    L.lzma_stream_buffer_bound.restype = C.c_size_t
    L.lzma_stream_buffer_bound.argtypes = [C.c_size_t]
    L.lzma_easy_buffer_encode.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(C.c_size_t), C.c_size_t]
    data = o.corpus_mixed(3 << 20, 12)
    bound = L.lzma_stream_buffer_bound(len(data))
    assert bound >= len(data) + 64
    out = C.create_string_buffer(bound + 7)
    pos = C.c_size_t(7)
    r = L.lzma_easy_buffer_encode(6, 4, None, data, len(data), out, C.byref(pos), len(out))
    assert r == 0 and pos.value > 7 + 64
    got = out.raw[7:pos.value]
    rr, dec = o.ref_decode(got, len(data) + 16) if o.have_ref() else (1, data)
    assert rr == 1 and dec == data
    opts = xz_amd.preset_options(6)
    want = o.orc_xz_stream(data, o.params_for_gpu_options(opts), xz_amd.mt_block_size(opts))
    assert o.first_diff(got, want) == -1
    small = C.create_string_buffer(1000)
    pos = C.c_size_t(3)
    assert L.lzma_easy_buffer_encode(6, 4, None, data, len(data), small, C.byref(pos), len(small)) == 10   # LZMA_BUF_ERROR
    assert pos.value == 3
    assert L.lzma_easy_buffer_encode(6, 4, None, data, len(data), None, C.byref(pos), 0) == 11            # LZMA_PROG_ERROR


def test_one_shot_single_block_layout(monkeypatch):
    """The one-shot API writes ONE Block whatever the input size, as the reference does (stream_buffer_encoder.c:91-101
    -> lzma_block_buffer_encode): 10 MiB at preset 1 (default Block size 3 MiB) with one span per Block is byte for byte
    the reference's lzma_easy_buffer_encode; 30 MiB at preset 6 (default Block size 24 MiB) is one Block, equal to the
    oracle's Stream at block_size = input size, and decodes through the reference."""
    import ctypes as C
    import xz_amd
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    L = xz_amd.lib()
    L.lzma_stream_buffer_bound.restype = C.c_size_t
    L.lzma_stream_buffer_bound.argtypes = [C.c_size_t]
    L.lzma_easy_buffer_encode.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.POINTER(C.c_size_t), C.c_size_t]

    def one_shot(data, preset):
        out = C.create_string_buffer(L.lzma_stream_buffer_bound(len(data)))
        pos = C.c_size_t(0)
        assert L.lzma_easy_buffer_encode(preset, 4, None, data, len(data), out, C.byref(pos), len(out)) == 0
        return out.raw[:pos.value]

    data = o.corpus_mixed(10 << 20, 31)
    monkeypatch.setenv("XZAMD_SPAN_KIB", str(64 << 10))          # one span per Block: the reference's own coding
    got = one_shot(data, 1)
    monkeypatch.delenv("XZAMD_SPAN_KIB")
    want = o.ref_easy_buffer_encode(data, 1)
    assert o.first_diff(got, want) == -1
    r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    assert r == 0 and nb == 1 and dec == data

    data = xz_amd.corpus_text(30 << 20, seed=77).tobytes()
    got = one_shot(data, 6)
    r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    assert r == 0 and nb == 1 and dec == data
    rr, rdec = o.ref_decode(got, len(data) + 16)
    assert rr == 1 and rdec == data
    opts = xz_amd.preset_options(6)
    want = o.orc_xz_stream(data, o.params_for_gpu_options(opts), len(data))
    assert o.first_diff(got, want) == -1


def test_lzma_code_semantics(tmp_path):
    """Option validation and action sequencing as in get_options (stream_encoder_mt.c:956-1000) and
    lzma_code (common/common.c:203-376), driven through ctypes."""
    import ctypes as C
    import xz_amd
    L = xz_amd.lib()

    class Stream(C.Structure):
        _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_size_t), ("total_in", C.c_uint64),
                    ("next_out", C.c_void_p), ("avail_out", C.c_size_t), ("total_out", C.c_uint64),
                    ("allocator", C.c_void_p), ("internal", C.c_void_p),
                    ("rp1", C.c_void_p), ("rp2", C.c_void_p), ("rp3", C.c_void_p), ("rp4", C.c_void_p),
                    ("seek_pos", C.c_uint64), ("ri2", C.c_uint64), ("ri3", C.c_size_t), ("ri4", C.c_size_t),
                    ("re1", C.c_int), ("re2", C.c_int)]

    class Mt(C.Structure):
        _fields_ = [("flags", C.c_uint32), ("threads", C.c_uint32), ("block_size", C.c_uint64),
                    ("timeout", C.c_uint32), ("preset", C.c_uint32), ("filters", C.c_void_p),
                    ("check", C.c_int), ("re1", C.c_int), ("re2", C.c_int), ("re3", C.c_int),
                    ("ri1", C.c_uint32), ("ri2", C.c_uint32), ("ri3", C.c_uint32), ("ri4", C.c_uint32),
                    ("memlimit_threading", C.c_uint64), ("memlimit_stop", C.c_uint64),
                    ("ri7", C.c_uint64), ("ri8", C.c_uint64),
                    ("rp1", C.c_void_p), ("rp2", C.c_void_p), ("rp3", C.c_void_p), ("rp4", C.c_void_p)]

    OK, STREAM_END, UNSUP, OPTIONS, BUF, PROG = 0, 1, 3, 8, 10, 11
    RUN, SYNC, FULL_FLUSH, FINISH, BARRIER = 0, 1, 2, 3, 4
    s = Stream()
    assert L.lzma_stream_encoder_mt(C.byref(s), None) == PROG
    for bad in (dict(threads=0), dict(threads=16385), dict(flags=1), dict(preset=10)):
        m = Mt(threads=2, preset=6, check=4)
        for k, v in bad.items():
            setattr(m, k, v)
        assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OPTIONS, bad
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(Mt(threads=1, preset=6, check=16))) == PROG
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(Mt(threads=1, preset=6, check=7))) == UNSUP
    L.lzma_stream_encoder_mt_memusage.restype = C.c_uint64
    assert L.lzma_stream_encoder_mt_memusage(C.byref(Mt(threads=0, preset=6))) == 2**64 - 1
    assert 0 < L.lzma_stream_encoder_mt_memusage(C.byref(Mt(threads=4, preset=6, check=4))) < 2**63

    m = Mt(threads=4, preset=1, check=4, block_size=1 << 16)
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OK
    data = o.corpus_lorem(300000)
    ib = C.create_string_buffer(data, len(data))
    ob = C.create_string_buffer(1 << 20)
    s.next_in = C.cast(ib, C.c_void_p).value
    s.avail_in = 100000
    s.next_out = C.cast(ob, C.c_void_p).value
    s.avail_out = len(ob)
    assert L.lzma_code(C.byref(s), SYNC) == PROG            # SYNC_FLUSH unsupported by the MT encoder
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OK   # re-init on the same strm
    assert L.lzma_code(C.byref(s), RUN) == OK and s.avail_in == 0
    assert L.lzma_code(C.byref(s), RUN) == OK               # first zero-progress call is tolerated
    assert L.lzma_code(C.byref(s), RUN) == BUF              # the second one is LZMA_BUF_ERROR
    # FULL_FLUSH ends the current Block early and returns STREAM_END, then RUN continues
    assert L.lzma_code(C.byref(s), FULL_FLUSH) == STREAM_END
    first_part = s.total_out
    assert first_part > 12
    s.avail_in = len(data) - 100000
    r = L.lzma_code(C.byref(s), FINISH)
    while r == OK:
        r = L.lzma_code(C.byref(s), FINISH)
    assert r == STREAM_END
    assert L.lzma_code(C.byref(s), FINISH) == STREAM_END     # sticky after the end
    pin, pout = C.c_uint64(0), C.c_uint64(0)
    L.lzma_get_progress(C.byref(s), C.byref(pin), C.byref(pout))
    assert pin.value == len(data) and pout.value == s.total_out
    got = ob.raw[: s.total_out]
    r2, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    assert r2 == 0 and dec == data
    assert nb == 2 + 4          # 100000 B -> 2 Blocks (flush), 200000 B -> 4 Blocks of <= 64 KiB
    # changing the action mid-sequence is a programming error
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OK
    s.next_in = C.cast(ib, C.c_void_p).value; s.avail_in = 10
    s.next_out = C.cast(ob, C.c_void_p).value; s.avail_out = 4      # too small: FINISH stays pending
    assert L.lzma_code(C.byref(s), FINISH) == OK
    assert L.lzma_code(C.byref(s), RUN) == PROG
    L.lzma_end(C.byref(s))
    assert not s.internal
    L.lzma_end(C.byref(s))      # idempotent


def test_progress_moves_inside_a_job():
    """lzma_get_progress (common/common.c:406, stream_encoder_mt.c:1004-1024; the reference's workers publish how far they
    are inside their Block, :261-267): progress_in moves while a job is on the GPU -- by the shares of its finished stages
    -- not only when a whole job is done, never goes back, never exceeds what was handed in, and ends at total_in."""
    import ctypes as C
    import sys
    import xz_amd
    if os.path.join(o.ROOT, "tools") not in sys.path:
        sys.path.insert(0, os.path.join(o.ROOT, "tools"))
    from bench_lzma_code import Mt, Stream
    L = xz_amd.lib()
    n = 192 << 20                                           # one job (< 1 GiB), 8 Blocks at preset 6
    data = xz_amd.corpus_text(n, seed=5)
    ob = C.create_string_buffer(n // 2)
    s = Stream()
    m = Mt(threads=1, preset=6, check=4, timeout=5)         # lzma_code returns every 5 ms while the worker is busy
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == 0
    s.next_in, s.avail_in = data.ctypes.data, n
    s.next_out, s.avail_out = C.cast(ob, C.c_void_p).value, len(ob)
    seen, last = set(), 0
    pin, pout = C.c_uint64(0), C.c_uint64(0)
    r = 0
    while r == 0:
        r = L.lzma_code(C.byref(s), 3)
        L.lzma_get_progress(C.byref(s), C.byref(pin), C.byref(pout))
        assert last <= pin.value <= n, (last, pin.value)
        last = pin.value
        seen.add(pin.value)
    assert r == 1 and s.total_in == n
    L.lzma_get_progress(C.byref(s), C.byref(pin), C.byref(pout))
    assert pin.value == n and pout.value == s.total_out
    inside = sorted(v for v in seen if 0 < v < n)
    assert len(inside) >= 2, sorted(seen)                   # e.g. 25 %, 40 %, 90 % of the job
    L.lzma_end(C.byref(s))
    rr, dec = o.ref_decode(ob.raw[: s.total_out], n + 16) if o.have_ref() else (1, data.tobytes())
    assert rr == 1 and dec == data.tobytes()


def test_lzma_code_worker_pipeline_timeout_barrier_and_filters_update(monkeypatch):
    """The front end deals batches of Blocks ("jobs") to one worker thread per GPU and drains them in order:
    many small jobs (XZAMD_BATCH_MIB=1) must give the same Stream as one big batch; lzma_mt.timeout makes a
    call return LZMA_OK while the workers are busy (stream_encoder_mt.c:667-713) without tripping the
    LZMA_BUF_ERROR rule; LZMA_FULL_BARRIER returns as soon as the input is handed over (:803-807);
    lzma_filters_update between Blocks changes the chain from the next Block on (:914-950); and the two
    helper exports answer (filter_encoder.c:270, hardware_cputhreads.c)."""
    import ctypes as C
    import xz_amd
    sys_path = os.path.join(o.ROOT, "tools")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from bench_lzma_code import Mt, Stream
    L = xz_amd.lib()
    OK, STREAM_END = 0, 1
    RUN, FULL_FLUSH, FINISH, BARRIER = 0, 2, 3, 4
    data = o.corpus_mixed(6 << 20, 33)
    ib = C.create_string_buffer(data, len(data))
    ob = C.create_string_buffer(len(data) + (1 << 20))

    def run(timeout, chunk):
        s = Stream()
        m = Mt(threads=4, preset=1, check=4, block_size=1 << 18, timeout=timeout)
        assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OK
        s.next_out = C.cast(ob, C.c_void_p).value
        s.avail_out = len(ob)
        pos = 0
        oks = 0
        while pos < len(data):
            k = min(chunk, len(data) - pos)
            s.next_in = C.cast(ib, C.c_void_p).value + pos
            s.avail_in = k
            while s.avail_in:
                assert L.lzma_code(C.byref(s), RUN) == OK
            pos += k
        r = L.lzma_code(C.byref(s), FINISH)
        while r == OK:
            oks += 1
            assert oks < 100000
            r = L.lzma_code(C.byref(s), FINISH)
        assert r == STREAM_END
        got = ob.raw[: s.total_out]
        L.lzma_end(C.byref(s))
        return got, oks

    monkeypatch.setenv("XZAMD_BATCH_MIB", "1")
    small, _ = run(0, 300000)
    timed, oks = run(1, 1 << 20)             # 1 ms: FINISH returns LZMA_OK (often without progress) until the jobs are done
    # several workers (two and three contexts on this GPU, XZAMD_TEST_WORKERS): jobs finish out of order and
    # are drained in order (stream_encoder_mt.c:599-665, outqueue.c:182-260) -- same Stream as with one worker
    L.xzamd_release_parked()
    monkeypatch.setenv("XZAMD_TEST_WORKERS", "2")
    two, _ = run(0, 300000)
    monkeypatch.setenv("XZAMD_TEST_WORKERS", "3")
    three, _ = run(2, 1 << 20)
    monkeypatch.delenv("XZAMD_TEST_WORKERS")
    L.xzamd_release_parked()
    assert two == small and three == small
    monkeypatch.delenv("XZAMD_BATCH_MIB")
    # several GPUs under LZMA_RUN (how `xz` feeds): jobs no larger than (input seen so far) / GPUs, so that every worker gets
    # work before the end of the input is known -- here with the 256 MiB floor turned down to 1 MiB: jobs of 1, 1, 1, 2, ... MiB
    monkeypatch.setenv("XZAMD_TEST_WORKERS", "2")
    monkeypatch.setenv("XZAMD_TEST_JOB_MIN_MIB", "1")
    grow, _ = run(0, 300000)
    monkeypatch.delenv("XZAMD_TEST_WORKERS")
    monkeypatch.delenv("XZAMD_TEST_JOB_MIN_MIB")
    L.xzamd_release_parked()
    assert grow == small
    big, _ = run(0, len(data))
    assert small == big and timed == big
    r, dec, nb = o.orc_xz_decode(big, len(data) + 16)
    assert r == 0 and dec == data and nb == 24

    # FULL_BARRIER: STREAM_END without waiting; the output arrives with the following calls
    s = Stream()
    m = Mt(threads=1, preset=1, check=4, block_size=1 << 18)
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OK
    s.next_in = C.cast(ib, C.c_void_p).value
    s.avail_in = 1 << 20
    s.next_out = C.cast(ob, C.c_void_p).value
    s.avail_out = len(ob)
    assert L.lzma_code(C.byref(s), BARRIER) == STREAM_END and s.avail_in == 0
    # new chain between Blocks: preset-0 options for the rest
    class Filter(C.Structure):
        _fields_ = [("id", C.c_uint64), ("options", C.c_void_p)]
    class OptLzma(C.Structure):
        _fields_ = [("dict_size", C.c_uint32), ("preset_dict", C.c_void_p), ("preset_dict_size", C.c_uint32),
                    ("lc", C.c_uint32), ("lp", C.c_uint32), ("pb", C.c_uint32), ("mode", C.c_int),
                    ("nice_len", C.c_uint32), ("mf", C.c_int), ("depth", C.c_uint32), ("pad", C.c_uint8 * 64)]
    ol = OptLzma(dict_size=1 << 18, lc=3, lp=0, pb=2, mode=1, nice_len=128, mf=3, depth=4)
    fl = (Filter * 2)(Filter(0x21, C.cast(C.pointer(ol), C.c_void_p)), Filter(2**64 - 1, None))
    assert L.lzma_filters_update(C.byref(s), fl) == OK
    s.next_in = C.cast(ib, C.c_void_p).value + (1 << 20)
    s.avail_in = 1 << 20
    r = L.lzma_code(C.byref(s), FINISH)
    while r == OK:
        r = L.lzma_code(C.byref(s), FINISH)
    assert r == STREAM_END
    got = ob.raw[: s.total_out]
    L.lzma_end(C.byref(s))
    r, dec, nb = o.orc_xz_decode(got, (2 << 20) + 16)
    assert r == 0 and dec == data[: 2 << 20] and nb == 8
    if o.have_ref():
        rr, rdec = o.ref_decode(got, (2 << 20) + 16)
        assert rr == 1 and rdec == data[: 2 << 20]
    # a chain in the middle of a Block is refused
    s = Stream()
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == OK
    s.next_in = C.cast(ib, C.c_void_p).value; s.avail_in = 1000
    s.next_out = C.cast(ob, C.c_void_p).value; s.avail_out = len(ob)
    assert L.lzma_code(C.byref(s), RUN) == OK
    assert L.lzma_filters_update(C.byref(s), fl) == 11
    L.lzma_end(C.byref(s))
    L.lzma_mt_block_size.restype = C.c_uint64
    assert L.lzma_mt_block_size(fl) == 1 << 20
    ol.dict_size = 8 << 20
    assert L.lzma_mt_block_size(fl) == 24 << 20
    assert L.lzma_cputhreads() >= 1
    # an option set outside the device path is refused at init (a preload client then stays on the CPU library)
    ol.dict_size = (1 << 30) + (1 << 28); ol.mode = 2; ol.mf = 0x14                 # dictionary > 1 GiB
    mm = Mt(threads=1, check=4, filters=C.cast(fl, C.c_void_p))
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(mm)) == 8
    # pb = 4 with LZMA_MODE_NORMAL runs on the device (two-phase mode), with a BT and with an HC finder
    for mf in (0x14, 0x04):
        ol.dict_size = 1 << 20; ol.pb = 4; ol.lc = 3; ol.lp = 0; ol.mode = 2; ol.mf = mf; ol.depth = 0
        assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(mm)) == 0
        s.next_in = C.cast(ib, C.c_void_p).value; s.avail_in = 700000
        s.next_out = C.cast(ob, C.c_void_p).value; s.avail_out = len(ob)
        r = L.lzma_code(C.byref(s), FINISH)
        while r == OK:
            r = L.lzma_code(C.byref(s), FINISH)
        assert r == STREAM_END
        got = ob.raw[: s.total_out]
        L.lzma_end(C.byref(s))
        if o.have_ref():
            rr, rdec = o.ref_decode(got, 700016)
            assert rr == 1 and rdec == data[:700000]
        r, dec, nb = o.orc_xz_decode(got, 700016)
        assert r == 0 and dec == data[:700000]
    ol.pb = 2; ol.lc = 4; ol.lp = 0; ol.mode = 1; ol.mf = 4; ol.depth = 8; ol.dict_size = 1 << 20
    assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(mm)) == 0       # lc + lp = 4 runs on the device
    s.next_in = C.cast(ib, C.c_void_p).value; s.avail_in = 500000
    s.next_out = C.cast(ob, C.c_void_p).value; s.avail_out = len(ob)
    r = L.lzma_code(C.byref(s), FINISH)
    while r == OK:
        r = L.lzma_code(C.byref(s), FINISH)
    assert r == STREAM_END
    got = ob.raw[: s.total_out]
    L.lzma_end(C.byref(s))
    r, dec, nb = o.orc_xz_decode(got, 500016)
    assert r == 0 and dec == data[:500000]
    L.xzamd_release_parked()
