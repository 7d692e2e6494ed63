"""Device .xz decoder against the REAL reference decoder (oracle/_ref) and the reference's own
fixture files: good-* files with an {LZMA2} chain decode to the same bytes, bad-* files are rejected, and
everything the device encoder emits decodes back -- Block-parallel and as a span-parallel verification decode."""
import glob
import os

import numpy as np
import pytest

import _oracle as o

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_files")


@pytest.fixture(scope="module")
def enc():
    import xz_amd
    e = xz_amd.Encoder()
    yield e
    e.close()


def _cuda(b):
    import torch
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() if len(b) else torch.empty(0, dtype=torch.uint8, device="cuda")


def test_reference_fixture_files(enc):
    import xz_amd
    good = sorted(glob.glob(os.path.join(FIX, "good-*.xz")))
    bad = sorted(glob.glob(os.path.join(FIX, "bad-*.xz")))
    assert len(good) >= 10 and len(bad) >= 10
    ndec = 0
    for f in good:
        raw = open(f, "rb").read()
        rr, want = o.ref_decode(raw, 1 << 20) if o.have_ref() else (1, None)
        if want is None:
            r, want, _ = o.orc_xz_decode(raw, 1 << 20)
            assert r == 0
        try:
            got, nb = enc.decode(_cuda(raw), 1 << 20)
        except xz_amd.XzAmdError as e:
            assert "(8)" in str(e), (f, str(e))      # chains other than {LZMA2}: declined, not mis-decoded
            continue
        assert got.cpu().numpy().tobytes() == want, f
        ndec += 1
    assert ndec >= 10
    accepted = []
    for f in bad:
        raw = open(f, "rb").read()
        try:
            enc.decode(_cuda(raw), 1 << 20)
            accepted.append(os.path.basename(f))
        except xz_amd.XzAmdError:
            pass
    assert accepted == [], accepted


@pytest.mark.parametrize("preset,bs,span", [(1, 1 << 18, 0), (6, 1 << 20, 0), (6, 100000, 8192), (3, 1 << 20, 0xFFFFFFFF),
                                            (9 | 0x80000000, 1 << 20, 0)])
def test_decodes_what_the_encoder_emits(enc, preset, bs, span):
    import torch
    import xz_amd
    data = o.corpus_mixed(1500000, 17) + o.corpus_random(70000) + o.corpus_lorem(300000)
    opts = xz_amd.preset_options(preset, span_size=span)
    t = _cuda(data)
    xz, _ = enc.encode(t, opts=opts, block_size=bs)
    xz = xz.clone()
    plain, nb = enc.decode(xz, len(data) + 16)
    assert plain.cpu().numpy().tobytes() == data and nb == (len(data) + bs - 1) // bs
    ver, nb2 = enc.decode(xz, len(data) + 16, expected=t)
    assert ver.cpu().numpy().tobytes() == data and nb2 == nb
    # corruption is noticed: a flipped payload byte, a wrong original
    bad = xz.clone()
    bad[xz.numel() // 2] ^= 0x40
    with pytest.raises(xz_amd.XzAmdError):
        enc.decode(bad, len(data) + 16)
    other = t.clone()
    other[len(data) // 3] ^= 1
    with pytest.raises(xz_amd.XzAmdError):
        enc.decode(xz, len(data) + 16, expected=other)


def test_decodes_reference_encoder_output(enc):
    """Streams of the REAL liblzma MT encoder (BT4, optimum_normal, 0x80 continuation chunks, uncompressed chunks)."""
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    data = o.corpus_mixed(900000, 5) + o.corpus_random(200000)
    for preset, bs in ((6, 1 << 18), (1, 0), (9, 300000)):
        raw = o.ref_encode_mt(data, preset, threads=2, block_size=bs)
        got, nb = enc.decode(_cuda(raw), len(data) + 16)
        assert got.cpu().numpy().tobytes() == data
        got2, _ = enc.decode(_cuda(raw), len(data) + 16, expected=_cuda(data))
        assert got2.cpu().numpy().tobytes() == data


def test_block_checks_are_verified_or_refused(enc):
    """Every Check the decoder accepts is verified (none, CRC32, CRC64, SHA-256: check/check.c, sha256.c); a Stream whose
    Check id it cannot verify is refused with XZAMD_UNSUPPORTED_CHECK (3) instead of being passed unverified; a
    verification decode against an original of another size is refused (no out-of-bounds reads of `expected`)."""
    import xz_amd
    data = o.corpus_mixed(700000, 3) + o.corpus_lorem(200000)
    t = _cuda(data)
    for check in (xz_amd.CHECK_NONE, xz_amd.CHECK_CRC32, xz_amd.CHECK_CRC64, xz_amd.CHECK_SHA256):
        for bs in (1 << 18, 250001):
            xz, _ = enc.encode(t, preset=1, block_size=bs, check=check)
            xz = xz.clone()
            got, nb = enc.decode(xz, len(data) + 16)
            assert got.cpu().numpy().tobytes() == data, (check, bs)
            if check != xz_amd.CHECK_NONE:
                # flip a bit of the last Block's stored Check (it sits right in front of the Index)
                raw = bytearray(xz.cpu().numpy().tobytes())
                isz = (int.from_bytes(raw[-8:-4], "little") + 1) * 4
                raw[len(raw) - 12 - isz - 1] ^= 0x10
                with pytest.raises(xz_amd.XzAmdError):
                    enc.decode(_cuda(bytes(raw)), len(data) + 16)
    if o.have_ref():
        raw = o.ref_encode_mt(data, 1, threads=2, block_size=1 << 18, check=10)       # SHA-256 from the real encoder
        got, _ = enc.decode(_cuda(raw), len(data) + 16)
        assert got.cpu().numpy().tobytes() == data
    # a Check id without a verifier: patch the Stream Flags of a CRC32 Stream to id 2 (same size) and fix their CRC32s
    xz, _ = enc.encode(t, preset=1, block_size=1 << 18, check=xz_amd.CHECK_CRC32)
    raw = bytearray(xz.cpu().numpy().tobytes())
    for off in (7, len(raw) - 3):
        raw[off] = 2
    import zlib
    raw[8:12] = zlib.crc32(bytes(raw[6:8])).to_bytes(4, "little")
    raw[-12:-8] = zlib.crc32(bytes(raw[-8:-2])).to_bytes(4, "little")
    with pytest.raises(xz_amd.XzAmdError) as ei:
        enc.decode(_cuda(bytes(raw)), len(data) + 16)
    assert "(3)" in str(ei.value), str(ei.value)          # XZAMD_UNSUPPORTED_CHECK
    xz, _ = enc.encode(t, preset=1, block_size=1 << 18)
    with pytest.raises(xz_amd.XzAmdError):
        enc.decode(xz.clone(), len(data) + 16, expected=t[: len(data) - 5].clone())
