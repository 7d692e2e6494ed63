"""Oracle decoder vs the reference's own fixture corpus (tests/files/README) and the real library."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as o

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MAN = json.load(open(os.path.join(GOLD, "manifest.json")))


@pytest.mark.parametrize("name", sorted(MAN["decode"]))
def test_reference_fixture(name):
    blob = open(os.path.join(GOLD, "ref_files", name), "rb").read()
    exp = MAN["decode"][name]
    r, dec, _ = o.orc_xz_decode(blob, 1 << 20)
    if exp["ok"]:
        assert r == 0, (name, r)
        assert len(dec) == exp["size"] and hashlib.sha256(dec).hexdigest() == exp["sha256"]
    else:
        assert r != 0, name


@pytest.mark.skipif(not o.have_ref(), reason="needs oracle/_ref")
@pytest.mark.parametrize("preset", [0, 1, 3, 4, 6])
def test_decodes_reference_encoder_output(preset):
    """Anything the real encoder emits (incl. BT4/normal presets) must decode with the oracle,
    and the oracle's parse trace must re-encode to the same bytes for fast presets."""
    for data in (o.corpus_lorem(300000), o.corpus_mixed(400000, 2), o.corpus_random(200000)):
        s = o.ref_encode_mt(data, preset, threads=2, block_size=1 << 18)
        r, dec, nb = o.orc_xz_decode(s, len(data) + 16)
        assert r == 0 and dec == bytes(data) and nb == (len(data) + (1 << 18) - 1) // (1 << 18)


def test_rejects_corruption():
    prm, _ = o.params_for_preset(1)
    data = o.corpus_lorem(50000)
    s = bytearray(o.orc_xz_stream(data, prm, 1 << 20))
    r, dec, _ = o.orc_xz_decode(bytes(s), len(data) + 16)
    assert r == 0 and dec == data
    rng = np.random.default_rng(0)
    hits = 0
    for _ in range(40):
        t = bytearray(s)
        i = int(rng.integers(0, len(t)))
        t[i] ^= 1 << int(rng.integers(0, 8))
        r, dec, _ = o.orc_xz_decode(bytes(t), len(data) + 16)
        hits += (r != 0)
    assert hits == 40   # every single-bit flip is caught by some CRC/size/grammar check
