"""x86 BCJ (simple/x86.c:26-118): the oracle restatement against the real reference filter."""
import numpy as np
import pytest

import _oracle as o

pytestmark = pytest.mark.skipif(not o.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("n,seed", [(0, 1), (1, 1), (4, 2), (5, 3), (6, 4), (9, 5), (100, 6), (5000, 7),
                                    (70000, 8), (300001, 9)])
def test_x86_filter_matches_reference(n, seed):
    data = o.corpus_x86(n, seed)
    if n == 0:
        assert o.orc_x86_encode(data) == b""
        return
    assert o.orc_x86_encode(data) == o.ref_x86_filter(data)


def test_x86_filter_edge_patterns():
    # opcodes in the last five bytes are not converted; back-to-back opcodes exercise prev_mask
    for data in (bytes([0xE8, 0, 0, 0, 0]), bytes([0xE8, 1, 2, 3, 0xFF, 0xE8]), bytes([0xE8] * 64),
                 bytes([0xE9, 0xE8, 0xE8, 0x00, 0xFF, 0x00, 0xE8, 0, 0, 0, 0, 0] * 20),
                 bytes([0x90] * 7 + [0xE8, 0x10, 0x00, 0x00, 0x00] + [0x90] * 3)):
        assert o.orc_x86_encode(data) == o.ref_x86_filter(data)


def test_x86_filter_changes_something_and_is_not_identity():
    data = o.corpus_x86(50000, 3)
    f = o.orc_x86_encode(data)
    assert f != data and len(f) == len(data)
    diff = np.nonzero(np.frombuffer(f, np.uint8) != np.frombuffer(data, np.uint8))[0]
    assert len(diff) > 100


def test_block_header_with_x86_matches_reference_stream():
    """Header/Index/framing of a {x86, LZMA2} Stream: re-frame the oracle's payload and compare with the
    reference MT encoder byte for byte (preset 1, one Block)."""
    data = o.corpus_x86(200000, 11)
    ref_stream = o.ref_encode_mt_x86(data, 1, threads=1, block_size=1 << 20)
    rr, dec = o.ref_decode(ref_stream, len(data) + 16)
    assert rr == 1 and dec == data
    # Block header: size byte, flags 0xC1 (two filters), ..., x86 filter flags 04 00 before LZMA2's 21 01 xx
    hdr = ref_stream[12:12 + (ref_stream[12] + 1) * 4]
    assert hdr[1] == 0xC1
    assert bytes([0x04, 0x00, 0x21, 0x01]) in hdr
