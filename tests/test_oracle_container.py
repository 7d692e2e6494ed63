"""Oracle container restatement vs the reference's own KATs and the real reference library.

KAT sources (paths relative to the XZ Utils tree):
  tests/test_vli.c:20-46      byte-exact VLI vectors
  tests/test_check.c:69-139   CRC32("123456789") = 0xCBF43926, CRC64 = 0x995DC9BBDF1939FA
  tests/test_stream_flags.c   header/footer magic + flags + CRC32 (checked via whole-stream equality)
"""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _oracle as o

needs_ref = pytest.mark.skipif(not o.have_ref(), reason="oracle/_ref (real reference build) not available")


def test_crc_kats():
    s = o.as_u8(b"123456789")
    assert o.orc().orc_crc32(o._ptr(s), 9, 0) == 0xCBF43926
    assert o.orc().orc_crc64(o._ptr(s), 9, 0) == 0x995DC9BBDF1939FA
    # incremental == one-shot (test_check.c rolling tests feed the data in pieces)
    c = 0
    for piece in (s[:1], s[1:4], s[4:]):
        c = o.orc().orc_crc64(o._ptr(np.ascontiguousarray(piece)), len(piece), c)
    assert c == 0x995DC9BBDF1939FA


def test_vli_kats():
    # tests/test_vli.c:20-46
    vecs = {0: b"\x00", 0x7F: b"\x7f", 0x80: b"\x80\x01", 0x3FFF: b"\xff\x7f", 0x4000: b"\x80\x80\x01",
            0x1FFFFF: b"\xff\xff\x7f", 0x200000: b"\x80\x80\x80\x01",
            0x7FFFFFFFFFFFFFFF: b"\xff" * 8 + b"\x7f"}
    buf = np.zeros(16, dtype=np.uint8)
    for v, want in vecs.items():
        n = o.orc().orc_vli_encode(v, o._ptr(buf))
        assert buf[:n].tobytes() == want, hex(v)


def test_crc64_combine_property():
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, size=100000, dtype=np.uint8)
    whole = o.orc().orc_crc64(o._ptr(data), len(data), 0)
    for cut in (0, 1, 4095, 4096, 50000, 99999, 100000):
        a = np.ascontiguousarray(data[:cut]); b = np.ascontiguousarray(data[cut:])
        ca = o.orc().orc_crc64(o._ptr(a), len(a), 0) if len(a) else 0
        cb = o.orc().orc_crc64(o._ptr(b), len(b), 0) if len(b) else 0
        assert o.orc().orc_crc64_combine(ca, cb, len(b)) == whole, cut


@needs_ref
def test_crc_vli_bound_vs_reference():
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 9, 255, 4096, 100001):
        d = rng.integers(0, 256, size=max(n, 1), dtype=np.uint8)
        assert o.orc().orc_crc32(o._ptr(d), n, 0) == o.ref().ref_crc32(o._ptr(d), n, 0)
        assert o.orc().orc_crc64(o._ptr(d), n, 0) == o.ref().ref_crc64(o._ptr(d), n, 0)
    a = np.zeros(16, dtype=np.uint8); b = np.zeros(16, dtype=np.uint8)
    for v in [0, 1, 127, 128, 300, 2**21 - 1, 2**21, 2**35 + 17, 2**63 - 1] + [int(x) for x in rng.integers(0, 2**62, size=50)]:
        n = o.orc().orc_vli_encode(v, o._ptr(a))
        m = C.c_size_t(0)
        assert o.ref().ref_vli_encode(v, o._ptr(b), 16, C.byref(m)) == 0
        assert a[:n].tobytes() == b[:m.value].tobytes()
    for u in (0, 1, 65535, 65536, 65537, 1 << 20, 3 << 20, 24 << 20, (192 << 20) + 5):
        assert o.orc().orc_block_bound(u) == o.ref().ref_block_buffer_bound(u)


@needs_ref
def test_uncompressed_block_writer_vs_reference():
    rng = np.random.default_rng(9)
    for n in (1, 100, 65536, 65537, 200000):
        d = rng.integers(0, 256, size=n, dtype=np.uint8)
        out = np.zeros(n + 4096, dtype=np.uint8); unp = C.c_uint64(0)
        k = o.orc().orc_block_uncomp_encode(o._ptr(d), n, 4, o._ptr(out), C.byref(unp))
        rout = np.zeros(n + 4096, dtype=np.uint8); rn = C.c_size_t(0); runp = C.c_uint64(0)
        assert o.ref().ref_block_uncomp_encode(o._ptr(d), n, 4, o._ptr(rout), len(rout), C.byref(rn), C.byref(runp)) == 0
        assert out[:k].tobytes() == rout[:rn.value].tobytes()
        assert unp.value == runp.value


@needs_ref
@pytest.mark.parametrize("preset", [0, 1, 2, 3])
def test_whole_stream_equals_reference_mt(preset):
    """Stream Header, Block Headers (sizes from maxima), padding, CRC64, Index, Footer: the oracle's
    framing of the oracle's payloads must equal lzma_stream_encoder_mt's output byte for byte."""
    data = o.corpus_lorem(229001)
    prm, _ = o.params_for_preset(preset)
    for bs in (65536, 1 << 20):
        mine = o.orc_xz_stream(data, prm, bs)
        theirs = o.ref_encode_mt(data, preset, threads=3, block_size=bs)
        assert o.first_diff(mine, theirs) == -1, (preset, bs)


@needs_ref
def test_incompressible_stream_equals_reference_mt():
    rng = np.random.default_rng(1)
    data = bytes(rng.integers(0, 256, size=300000, dtype=np.uint8))
    prm, _ = o.params_for_preset(1)
    for bs in (65536, 1 << 20):
        assert o.orc_xz_stream(data, prm, bs) == o.ref_encode_mt(data, 1, threads=2, block_size=bs)


def test_golden_mt_stream_hash():
    """Pinned without oracle/_ref: sha256 of the reference MT output recorded in manifest.json."""
    import json, os
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")))
    prm, _ = o.params_for_preset(1)
    for cname, data in (("text", o.corpus_lorem(229001)), ("abc", o.corpus_abc()), ("random", o.corpus_random())):
        assert hashlib.sha256(data).hexdigest() == man["encode"][cname]["sha256"]
        s = o.orc_xz_stream(data, prm, 65536)
        assert hashlib.sha256(s).hexdigest() == man["encode"][cname]["mt_preset1_bs64k"]["sha256"], cname


@needs_ref
def test_bench_whole_job_reference_child():
    """bench.py's `ratio.whole_job` leg: the child that runs the reference MT encoder over the whole (seeded) text corpus and
    prints its size -- no GPU involved; the size must be what the reference gives for the same bytes in this process."""
    import json
    import os
    import subprocess
    import sys
    import xz_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ref-size-child", "2", "--size-mib", "6", "--preset", "1"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    data = xz_amd.corpus_text(6 << 20, seed=1000)
    want = o.ref_encode_mt(data, 1, threads=2, block_size=xz_amd.mt_block_size(xz_amd.preset_options(1)))
    assert d["rc"] == 1 and d["in_bytes"] == 6 << 20 and d["ref_bytes"] == len(want)
