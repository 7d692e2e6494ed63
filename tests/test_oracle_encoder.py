"""Oracle fast-mode encoder: byte-identical to the real reference raw LZMA2 encoder (presets 0-3),
pinned additionally by recorded hashes; span mode decodes with both decoders."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as o

MAN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")))
needs_ref = pytest.mark.skipif(not o.have_ref(), reason="needs oracle/_ref")

CORPORA = {
    "text": lambda: o.corpus_lorem(229001),       # tests/create_compress_files.c:110-152
    "abc": lambda: o.corpus_abc(),                # :82-88
    "random": lambda: o.corpus_random(),          # :93-105
}


@pytest.mark.parametrize("cname", sorted(CORPORA))
@pytest.mark.parametrize("preset", [0, 1, 2, 3])
def test_golden_hash(cname, preset):
    data = CORPORA[cname]()
    prm, _ = o.params_for_preset(preset)
    enc = o.orc_encode_block(data, prm)
    exp = MAN["encode"][cname]["raw_lzma2"][str(preset)]
    assert len(enc) == exp["size"] and hashlib.sha256(enc).hexdigest() == exp["sha256"]


OWN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "own_definition.json")))


@pytest.mark.parametrize("name", sorted(OWN))
def test_own_definition_golden(name):
    """Presets 4-9 in product mode are OUR definition (finder, parser, piece / encode-span plan, two-phase coder): the
    reference's tests hold no vector for them.  The committed hashes of the oracle's output on small seeded inputs
    (tests/golden/make_own_golden.py) pin that definition against unintended edits and compiler / platform differences;
    the GPU tests pin the HIP path to the oracle byte for byte.  Every vector decodes through the oracle decoder (and the
    reference decoder where oracle/_ref is built)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_own_golden", os.path.join(os.path.dirname(__file__), "golden", "make_own_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    data, preset, over = mod.cases()[name]
    exp = OWN[name]
    if not (len(data) == exp["in_size"] and hashlib.sha256(data).hexdigest() == exp["in_sha256"]):
        pytest.skip("the seeded input generators give other bytes here (another numpy?): the vector pins the encoder, not them")
    raw = mod.encode(data, preset, over)
    assert len(raw) == exp["size"] and hashlib.sha256(raw).hexdigest() == exp["sha256"], (name, len(raw), exp["size"])
    import xz_amd
    dict_size = xz_amd.preset_options(preset).dict_size
    r2, dec2 = o.orc_decode_raw(raw, dict_size, len(data) + 16)
    assert r2 == 0 and dec2 == data
    if o.have_ref():
        r, dec = o.ref_raw_decode(raw, dict_size, len(data) + 16)
        assert r == 1 and dec == data


def _bench_text(n):
    """First bytes of bench.py's corpus (xzamd_corpus_text is plain host code of the product library; used
    here only as a data generator)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import xz_amd
    return xz_amd.corpus_text(max(n, 1 << 16), seed=1000).tobytes()[:n]


def _edge_inputs():
    rng = np.random.default_rng(21)
    lorem = o.corpus_lorem(1 << 20)
    rnd = bytes(rng.integers(0, 256, size=300000, dtype=np.uint8))
    return {
        "empty": b"", "one": b"x", "two": b"xy", "three": b"aaa", "four": b"abcd", "run": b"\0" * 100000,
        "period3": b"abc" * 30000, "lorem1M": lorem, "mixed": o.corpus_mixed(700000, 5),
        "rnd": rnd, "sandwich": lorem[:150000] + rnd[:200000] + lorem[:100000],
        "long_match": lorem[:5000] * 40, "binaryish": bytes(rng.integers(0, 4, size=200000, dtype=np.uint8)),
        "x86ish": o.corpus_x86(400000, 31), "x86_filtered": o.orc_x86_encode(o.corpus_x86(300000, 32)),
        "enwik_style": _bench_text(600000),
    }


@needs_ref
@pytest.mark.parametrize("preset", [0, 1, 2, 3])
def test_identical_to_reference_raw_encoder(preset):
    prm, _ = o.params_for_preset(preset)
    for name, data in _edge_inputs().items():
        mine = o.orc_encode_block(data, prm)
        theirs = o.ref_raw_encode(data, prm, mode=1)
        assert o.first_diff(mine, theirs) == -1, (name, preset)


@needs_ref
def test_identical_with_custom_options():
    """Non-preset corners: tiny dictionary (chain cut by cyclic_size), lc/lp/pb variants, depth 1."""
    data = o.corpus_mixed(500000, 8)
    for dict_size, lc, lp, pb, nice, mf, depth in [
            (4096, 3, 0, 2, 32, 4, 0), (65536, 0, 2, 0, 273, 4, 1), (1 << 20, 4, 0, 4, 8, 3, 0),
            (12345, 2, 2, 1, 64, 4, 100), (1 << 16, 3, 0, 2, 32, 3, 0)]:
        p = o.OrcParams(dict_size, lc, lp, pb, nice, mf, depth, 0)
        assert o.orc_encode_block(data, p) == o.ref_raw_encode(data, p, mode=1), (dict_size, lc, lp, pb, nice, mf, depth)


@pytest.mark.parametrize("span", [4096, 65536, 100000])
def test_span_mode_roundtrip(span):
    for name, data in _edge_inputs().items():
        prm, _ = o.params_for_preset(1, span_size=span)
        enc = o.orc_encode_block(data, prm)
        r, dec, syms, tr = o.orc_decode_raw(enc, prm.dict_size, len(data) + 16, want_trace=True)
        assert r == 0 and dec == bytes(data), (name, span)
        if o.have_ref():
            out = np.empty(len(data) + 16, dtype=np.uint8); n = C.c_size_t(0)
            pl = o.as_u8(enc)
            rr = o.ref().ref_raw_lzma2_decode(o._ptr(pl), len(pl), prm.dict_size, o._ptr(out), len(out), C.byref(n))
            assert rr == 1 and out[:n.value].tobytes() == bytes(data), (name, span)


def test_span_cost_is_small():
    data = o.corpus_lorem(4 << 20)
    prm0, _ = o.params_for_preset(1)
    base = len(o.orc_encode_block(data, prm0))
    prm, _ = o.params_for_preset(1, span_size=65536)
    assert len(o.orc_encode_block(data, prm)) <= base * 1.03


def test_parse_trace_consistency():
    """Decoder-extracted parse == encoder-emitted parse (ties the two restatements together)."""
    data = o.corpus_mixed(300000, 4)
    prm, _ = o.params_for_preset(2)
    enc, esym, _ = o.orc_encode_block(data, prm, want_trace=True)
    r, dec, dsym, _ = o.orc_decode_raw(enc, prm.dict_size, len(data) + 16, want_trace=True)
    assert r == 0 and dec == data
    assert esym.shape == dsym.shape and (esym == dsym).all()


# ---- OUR definitions (GPU successor of BT4 + windowed optimal parser): CPU-side sanity ----------
@pytest.mark.parametrize("depth2,parser,span", [(5, 1, 0), (0, 1, 65536), (5, 1, 131072), (3, 1, 4096), (1, 1, 8192)])
def test_sn_finder_and_optimal_parser_roundtrip(depth2, parser, span):
    for name, data in _edge_inputs().items():
        if len(data) > 400000:
            data = data[:400000]
        p = o.OrcParams(1 << 20, 3, 0, 2, 64, 4, 8 if depth2 else 24, span, depth2, parser)
        enc = o.orc_encode_block(data, p)
        r, dec = o.orc_decode_raw(enc, p.dict_size, len(data) + 16)
        assert r == 0 and dec == bytes(data), (name, depth2, parser, span)
        if o.have_ref():
            out = np.empty(len(data) + 16, dtype=np.uint8); n = C.c_size_t(0)
            pl = o.as_u8(enc)
            rr = o.ref().ref_raw_lzma2_decode(o._ptr(pl), len(pl), p.dict_size, o._ptr(out), len(out), C.byref(n))
            assert rr == 1 and out[:n.value].tobytes() == bytes(data), (name, depth2, parser, span)


def test_optimal_parser_beats_fast_parser():
    """The point of the suffix-neighbourhood finder and the windowed optimal parser: smaller output than the
    reference's fast mode, and close to its normal mode (the tolerance the GPU tests enforce at size)."""
    data = o.corpus_lorem(1 << 20)
    base = len(o.orc_encode_block(data, o.OrcParams(1 << 23, 3, 0, 2, 273, 4, 56, 0, 0, 0)))
    exact_opt = len(o.orc_encode_block(data, o.OrcParams(1 << 23, 3, 0, 2, 64, 4, 8, 0, 0, 1)))
    opt = len(o.orc_encode_block(data, o.OrcParams(1 << 23, 3, 0, 2, 64, 4, 1, 0, 5, 1)))
    assert opt < exact_opt and opt < base
    if o.have_ref():
        prm = o.OrcParams(1 << 23, 3, 0, 2, 64, 0x14, 0, 0, 0, 0)
        ref6 = len(o.ref_raw_encode(data, prm, mode=2))      # liblzma preset-6 options (BT4, normal)
        assert opt <= ref6 * (1 + SIZE_TOLERANCE), (opt, ref6)


SIZE_TOLERANCE = 0.03      # the stated tolerance (presets 4-9, FULL Blocks through the product path: 2.5 %, tests/test_gpu_parity.py) with the
                           # margin the 4 ... 6 MiB Blocks of this CPU twin need: a Block's first MiBs are where the model is least trained
                           # (RGBA pixels: +2.58 % on 4 MiB, +2.25 % on 24 MiB)
SIZE_TOLERANCE_FAST = 0.01 # presets 1-3 with the default 256 KiB spans


def _elf_mix(n):
    """x86-64 shared objects of the image, 1 MiB from each, in sorted order (present here and on the GPU box)."""
    import glob, os
    files = sorted(glob.glob("/usr/lib/x86_64-linux-gnu/*.so*"))
    files = [f for f in files if os.path.isfile(f) and not os.path.islink(f) and os.path.getsize(f) > 65536]
    out = bytearray()
    for f in files:
        with open(f, "rb") as fh:
            out += fh.read(1 << 20)
        if len(out) >= n:
            break
    return bytes(out[:n])


@pytest.mark.parametrize("corpus,n", [("lorem", 4 << 20), ("text", 4 << 20), ("elf", 4 << 20), ("tar", 6 << 20),
                                      ("rocm_headers", 6 << 20), ("logs", 4 << 20), ("json", 4 << 20),
                                      ("sqlite", 4 << 20), ("dpkg_tar", 6 << 20),
                                      # round 5: the literal-heavy / numeric classes of the round-4 review
                                      ("f32sine", 4 << 20), ("f32two", 4 << 20), ("f32mesh", 4 << 20), ("fasta", 4 << 20),
                                      ("sparse", 4 << 20), ("html", 4 << 20), ("csv", 4 << 20), ("pcm16", 4 << 20),
                                      ("f64sine", 4 << 20), ("int32walk", 4 << 20), ("structs24", 4 << 20), ("hexids", 4 << 20),
                                      # round 6: the classes the round-5 review probed and found outside
                                      ("rgba", 4 << 20), ("varint", 4 << 20), ("cjk", 4 << 20), ("cycled_tree", 6 << 20)])
def test_size_within_tolerance_of_reference_preset6(corpus, n):
    """Oracle restatement of what the device runs for preset 6 (64-byte suffix order, cost-balanced spans) against
    the REAL liblzma at preset 6 on the same Block: compressed size within the stated tolerance.  (The GPU test
    repeats this on full 24 MiB Blocks through the product path.)"""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    import xz_amd
    if corpus == "lorem":
        data = o.corpus_lorem(n)
    elif corpus == "text":
        data = xz_amd.corpus_text(n, seed=1000).tobytes()
    elif corpus in ("tar", "rocm_headers"):
        try:
            data = (xz_amd.corpus_tar(n) if corpus == "tar" else xz_amd.corpus_tar(n, "/opt/rocm/include")).tobytes()
        except RuntimeError:
            pytest.skip("the image has none of the source trees the tar corpus is made of")
    elif corpus in ("logs", "json", "sqlite", "dpkg_tar"):
        import _corpora
        data = {"logs": _corpora.logs, "json": _corpora.json_records, "sqlite": _corpora.sqlite_file,
                "dpkg_tar": _corpora.dpkg_tar}[corpus](n)
        if data is None:
            pytest.skip("no /var/lib/dpkg on this box")
    elif corpus == "elf":
        data = _elf_mix(n)
        if len(data) < n:
            pytest.skip("not enough ELF files on this box")
    else:
        import _corpora
        data = (_corpora.REVIEW_CLASSES[corpus] if corpus in _corpora.REVIEW_CLASSES else _corpora.NUMERIC_CLASSES[corpus])(n)
        if data is None:
            pytest.skip("class not available on this image")
    prm = o.params_for_gpu_options(xz_amd.preset_options(6))
    assert prm.span_cost and prm.sa_depth == 64 and prm.enc_bits        # two-phase: parse pieces + encode spans
    ours_raw = o.orc_encode_block(data, prm)
    r, dec = o.ref_raw_decode(ours_raw, prm.dict_size, len(data) + 16)
    assert r == 1 and dec == data
    ref = len(o.ref_raw_encode(data, o.OrcParams(1 << 23, 3, 0, 2, 64, 0x14, 0, 0, 0, 0), mode=2))
    # (varint records: +3.3 % on 4 MiB, +2.7 % on 8 MiB, +1.7 % on the full 24 MiB Block the GPU test holds to 2.5 %)
    tol = {"varint": 0.04}.get(corpus, SIZE_TOLERANCE)
    assert len(ours_raw) <= ref * (1 + tol), (corpus, len(ours_raw), ref, len(ours_raw) / ref - 1)


def test_size_distribution_over_random_record_tables():
    """CPU twin of the GPU test of the same name (there: full 24 MiB Blocks through the product path): 32 draws of the seeded
    random class generator, 3 MiB each, oracle restatement vs the real liblzma at preset 6.  Pinned: the median and the share of
    the draws inside the stated tolerance; the tails (a table of counters: -77 %; constants + a counter + an enum: +57 %) are
    what DESIGN.md section 5 reports."""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    import concurrent.futures as cf
    import xz_amd
    import _corpora
    prm = o.params_for_gpu_options(xz_amd.preset_options(6))
    refp = o.OrcParams(1 << 23, 3, 0, 2, 64, 0x14, 0, 0, 0, 0)
    n = 3 << 20

    def one(seed):
        d = _corpora.random_class(seed, n)
        raw = o.orc_encode_block(d, prm)
        r, dec = o.ref_raw_decode(raw, prm.dict_size, n + 16)
        assert r == 1 and dec == d, seed
        return 100.0 * (len(raw) / len(o.ref_raw_encode(d, refp, mode=2)) - 1)
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        v = np.array(list(pool.map(one, range(32))))
    inside = float((v <= 100.0 * SIZE_TOLERANCE).mean())
    print("random record tables, 3 MiB, preset 6:", np.round(v, 2).tolist(), "median", float(np.median(v)), "inside", inside)
    assert float(np.median(v)) <= 1.0 and inside >= 0.70, np.round(v, 2).tolist()


def test_span_plan_properties():
    """Cost-balanced spans (oracle plan_spans): the spans tile the Block, start on 4 KiB boundaries, respect the
    minimum length (all but the last), adapt to the data (long spans over highly compressible stretches), and a
    larger work target gives fewer spans; the plan of an incompressible Block falls back to the minimum length."""
    import xz_amd
    rng = np.random.default_rng(3)
    lorem = o.corpus_lorem(1 << 20)
    data = lorem + b"\0" * (3 << 20) + bytes(rng.integers(0, 256, size=1 << 20, dtype=np.uint8)) + lorem[:500000]
    prm = o.params_for_gpu_options(xz_amd.preset_options(6))
    work, bits, starts = o.orc_span_plan(data, prm)
    assert starts[0] == 0 and (starts % 4096 == 0).all() and (np.diff(starts) > 0).all()
    lens = np.diff(np.append(starts, len(data)))
    assert (lens[:-1] >= prm.span_size).all()
    zero_spans = [l for s, l in zip(starts, lens) if (1 << 20) <= s and s + l <= (4 << 20)]
    # the run of zeros costs next to nothing: its pieces are as long as a piece may be -- 1 MiB (round 6: the round-5 plan let a
    # piece grow to 16 MiB, one wavefront walking it while thousands idle: config C5 lost a third of its throughput)
    assert max(lens) == 1 << 20 and len(zero_spans) <= 3
    rnd = [l for s, l in zip(starts, lens) if s >= (4 << 20) + 65536 and s + l <= (5 << 20)]
    assert rnd and max(rnd) <= 262144                               # incompressible bytes: one unit of work each
    prm2 = o.params_for_gpu_options(xz_amd.preset_options(6))
    prm2.span_cost = 4 * prm.span_cost
    _, _, starts2 = o.orc_span_plan(data, prm2)
    assert len(starts2) < len(starts)
    assert int(work.sum()) >= (len(data) - (3 << 20)) // 2
    # highly compressible Blocks: below one estimated bit per planned byte a piece must produce more than span_bits (up
    # to 4x): a piece start costs 150 ... 280 bytes whatever the data
    import _corpora
    sp = _corpora.sparse_text(8 << 20)
    prm3 = o.params_for_gpu_options(xz_amd.preset_options(6))
    bits3, starts3 = o.orc_span_plan(sp, prm3)[1], o.orc_piece_plan(sp, prm3)[0]
    planned_bits = int(bits3[16:].sum())                      # without the seed piece's 16 chunks
    assert planned_bits * 2 < len(sp) - 65536                    # < 0.5 estimated bits per byte
    lens3 = np.diff(np.append(starts3, len(sp)))
    assert lens3.max() <= 1 << 20                                # whatever the bit rule says: no piece longer than 1 MiB
    # the bit bound is at least twice as tight as span_bits alone (where the 1 MiB cap does not cut first)
    assert 1 + 1 <= len(starts3) - 1 <= max(planned_bits // prm3.span_bits // 2, (len(sp) - 65536 + (1 << 20) - 1) >> 20)
    txt = xz_amd.corpus_text(4 << 20, seed=2).tobytes()
    _, bits4, _ = o.orc_span_plan(txt, prm3)
    assert int(bits4.sum()) > len(txt)                           # text: > 1 estimated bit per byte, the plain span_bits bound


def test_tar_corpus_is_a_deterministic_ustar_stream(tmp_path):
    """xzamd_corpus_tar (BASELINE config C4's input): the same bytes every time, a stream `tar` itself accepts, cycled
    with a perturbation once the roots are exhausted (so that cycles are not identical)."""
    import subprocess
    import xz_amd
    root = tmp_path / "tree"
    (root / "b").mkdir(parents=True)
    (root / "a.txt").write_bytes(b"alpha\n" * 1000)
    (root / "b" / "c.h").write_bytes(b"#define X 1\n" * 700)
    (root / "b" / ("long_name_" * 12 + ".py")).write_bytes(b"print('x')\n" * 500)
    n = 200000
    a = xz_amd.corpus_tar(n, str(root), seed=7).tobytes()
    assert a == xz_amd.corpus_tar(n, str(root), seed=7).tobytes()
    first_cycle = 512 * 3 + 6144 + 8704 + 5632          # three headers + the bodies padded to 512
    assert a[:first_cycle] == xz_amd.corpus_tar(first_cycle, str(root), seed=8).tobytes()     # the first cycle is unperturbed
    assert a[first_cycle:2 * first_cycle] != a[:first_cycle]                                  # later cycles differ
    t = tmp_path / "c.tar"
    t.write_bytes(a[:first_cycle] + b"\0" * 1024)
    names = subprocess.run(["tar", "tf", str(t)], capture_output=True, text=True)
    assert names.returncode == 0 and len(names.stdout.split()) == 3, (names.stdout, names.stderr)
    with pytest.raises(RuntimeError):
        xz_amd.corpus_tar(1000, str(tmp_path / "missing"))


def test_two_phase_round_trip_many_small_pieces():
    """Two-phase semantics under stress (oracle level; the device is pinned to the oracle byte for byte by the GPU tests):
    4 KiB parse pieces -- thousands of piece starts, each with a pre-roll whose rep distances and coder state are the
    parser's guess, not what the coder has there -- and small encode spans.  The coder takes every literal from its
    record (byte, context byte, match byte when the record may carry one), exactly as k_encode_syms does; a record that
    carried a match byte across a piece start would decode to different bytes here (found at 10,000 pieces on the GPU)."""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    import xz_amd
    rng = np.random.default_rng(11)
    data = (xz_amd.corpus_text(3 << 20, seed=5).tobytes() + o.corpus_lorem(1 << 20) + xz_amd.corpus_tar(2 << 20).tobytes()
            + bytes(rng.integers(0, 4, size=300000, dtype=np.uint8)) + o.corpus_lorem(700000)[::-1])
    prm = o.params_for_gpu_options(xz_amd.preset_options(6))
    prm.span_size = 4096            # shortest piece (the device: 64 KiB)
    prm.span_cost = 4096
    prm.span_bits = 0
    prm.enc_bits = 200000
    starts, estarts = o.orc_piece_plan(data, prm)
    assert len(starts) > 700 and len(estarts) > 4
    raw = o.orc_encode_block(data, prm)
    r, dec = o.ref_raw_decode(raw, prm.dict_size, len(data) + 16)
    assert r == 1 and dec == data


@pytest.mark.parametrize("pb", [3, 4])
def test_two_phase_pb_above_two(pb):
    """pb = 3, 4 with the optimal parser (lzma/lzma_common.h:32-37): the parse pieces price with a pb = 2 view of the
    positions (their model is the parser's alone; the device parser's per-window tables hold four position states), the
    coder's continuous model runs the real pb.  The stream carries the real pb in its props byte, decodes through the
    reference decoder, and stays inside the stated size tolerance of liblzma at the SAME pb (BT4, normal mode) on 16-byte
    records -- the data pb = 4 is for (measured: pb 3 +1.4 %, pb 4 +1.8 %; pb 2 +1.9 %)."""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    import xz_amd
    rng = np.random.default_rng(3)
    k = (4 << 20) // 16
    rec = np.zeros((k, 16), dtype=np.uint8)
    rec[:, 0:4] = np.arange(k, dtype=np.uint32).view(np.uint8).reshape(k, 4)
    rec[:, 4:8] = (np.sin(np.arange(k) * 7e-4) * 50).astype(np.float32).view(np.uint8).reshape(k, 4)
    rec[:, 8:10] = rng.integers(0, 4, size=(k, 2), dtype=np.uint8)
    rec[:, 10:16] = rng.integers(0, 256, size=(k, 6), dtype=np.uint8) // 64
    data = rec.tobytes()
    opts = xz_amd.preset_options(6)
    opts.pb = pb
    prm = o.params_for_gpu_options(opts)
    assert prm.pb == pb and prm.enc_bits and prm.parser == 1
    raw = o.orc_encode_block(data, prm)
    assert raw[0] >> 5 == 7 and raw[5] == (pb * 5 + 0) * 9 + 3          # first chunk: dict reset + props (lzma2_encoder.c:54-107)
    r, dec = o.ref_raw_decode(raw, prm.dict_size, len(data) + 16)
    assert r == 1 and dec == data
    p2 = o.OrcParams()
    p2.dict_size, p2.lc, p2.lp, p2.pb, p2.nice_len, p2.mf, p2.depth = prm.dict_size, 3, 0, pb, 64, 0x14, 0
    ref = o.ref_raw_encode(data, p2, mode=2)
    assert len(raw) <= len(ref) * (1 + SIZE_TOLERANCE), (pb, len(raw), len(ref))
    prm2 = o.params_for_gpu_options(xz_amd.preset_options(6))
    assert len(raw) < len(o.orc_encode_block(data, prm2))              # the finer position contexts pay on such data


def test_two_phase_token_budget_overflow_goes_raw():
    """Round-4 advisor (medium): the model pass of the two-phase coder has a fixed token budget per encode span (10 per input
    byte); data made of far three-byte matches needs more.  Out of tokens, the chunk is closed where it stands and the rest
    of the span is stored as raw chunks -- a valid Stream, never an error.  Reached here on ordinary data with the budget
    turned down (the device: XZAMD_TEST_TOK_PER_BYTE, same rule, byte-identical: test_gpu_parity)."""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    import xz_amd
    rng = np.random.default_rng(5)
    data = (xz_amd.corpus_text(2 << 20, seed=9).tobytes() + bytes(rng.integers(0, 256, size=200000, dtype=np.uint8))
            + o.corpus_lorem(1 << 20))
    prm = o.params_for_gpu_options(xz_amd.preset_options(6))
    prm.enc_bits = 400000
    full = o.orc_encode_block(data, prm)
    try:
        sizes = {}
        for budget in (1, 2, 3):
            o.orc_set_tok_per_byte(budget)
            raw = o.orc_encode_block(data, prm)
            r, dec = o.ref_raw_decode(raw, prm.dict_size, len(data) + 16)
            assert r == 1 and dec == data, budget
            sizes[budget] = len(raw)
            r2, dec2, _, tr = o.orc_decode_raw(raw, prm.dict_size, len(data) + 16, want_trace=True)
            assert r2 == 0 and dec2 == data and tr.chunks_uncompressed > 3, budget       # the raw tails of the encode spans
    finally:
        o.orc_set_tok_per_byte(0)
    assert len(full) < sizes[3] < sizes[2] < sizes[1] < len(data) + len(data) // 1000
    assert o.orc_encode_block(data, prm) == full                 # the knob is off again


@pytest.mark.parametrize("corpus", ["text", "html", "lorem"])
def test_size_within_tolerance_of_reference_preset1(corpus):
    """Presets 1-3 in product mode = the exact HC4 finder and optimum_fast with a state reset every 256 KiB (the oracle
    restatement of what the device runs) against the REAL liblzma at preset 1 on one 6 MiB Block."""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    import xz_amd
    import _corpora
    n = 6 << 20
    data = {"text": lambda: xz_amd.corpus_text(n, seed=1000).tobytes(), "html": lambda: _corpora.html_rows(n),
            "lorem": lambda: o.corpus_lorem(n)}[corpus]()
    prm = o.params_for_gpu_options(xz_amd.preset_options(1))
    assert prm.span_size == 262144 and prm.parser == 0
    ours = o.orc_xz_stream(data, prm, n)
    ref = o.ref_encode_mt(data, 1, 2, n)
    assert len(ours) <= len(ref) * (1 + SIZE_TOLERANCE_FAST), (corpus, len(ours), len(ref))
