#!/usr/bin/env python3
"""Regenerates tests/golden/own_definition.json: sha256 of the ORACLE's output for presets 4-9 in product mode (the
suffix-neighbourhood finder, the windowed optimal parser, the piece / encode-span plan and the two-phase coder are OUR
definitions: the reference's tests hold no vector for them).  The vectors pin that definition against
unintended edits of oracle/lzma_fast_enc.c and against compiler / platform differences; the GPU tests pin the HIP path to
the oracle byte for byte, so they pin the product's bytes as well.  A deliberate change of the definition (round 6: carried
encode spans, two parse iterations, stored pieces, 16,000-byte chunks) regenerates this file -- say so in the commit.

    python tests/golden/make_own_golden.py          (needs only oracle/liboracle.so and xz_amd/libxz_amd.so's corpus generators)"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import _oracle as o  # noqa: E402


def cases():
    """name -> (data, preset, option overrides): small seeded inputs, each a few seconds of oracle time."""
    import xz_amd
    import _corpora
    rng = np.random.default_rng(2026)
    lorem = o.corpus_lorem(229001)                                   # tests/create_compress_files.c:110-152
    mix = (xz_amd.corpus_text(1 << 20, seed=3).tobytes() + b"\0" * 700000 + bytes(rng.integers(0, 256, size=200000, dtype=np.uint8))
           + lorem[:3000] * 150 + _corpora.random_class(5, 600000))      # (integer arithmetic only: no libm in the inputs)
    k = 40000
    rec = np.zeros((k, 16), dtype=np.uint8)
    rec[:, 0:4] = np.arange(k, dtype=np.uint32).view(np.uint8).reshape(k, 4)
    rec[:, 4:8] = (np.abs((np.arange(k) * 7) % 2000 - 1000) * 3).astype(np.uint32).view(np.uint8).reshape(k, 4)   # a triangle wave
    rec[:, 8:16] = rng.integers(0, 4, size=(k, 8), dtype=np.uint8)
    return {
        "lorem_p4": (lorem, 4, {}),
        "lorem_p6": (lorem, 6, {}),
        "lorem_p9e": (lorem, 9 | 0x80000000, {}),
        "mix_p6": (mix, 6, {}),
        "mix_p6_small_pieces": (mix, 6, {"span_cost": 40000, "span_bits": 50000, "enc_span_bits": 300000}),
        "sparse_p6": (_corpora.sparse_text(4 << 20), 6, {}),          # < 1 estimated bit per byte: the scaled bit bound of a piece
        "rec16_pb4_p6": (rec.tobytes(), 6, {"pb": 4}),                # pb = 4: the parser's pb = 2 view, the coder's real pb
        # round 6: image pixels (the carried coder model, snapshots), a table of records with three partial iterations, and
        # incompressible pieces inside compressible data (stored per piece by the parser's price)
        "rgba_p6": (_corpora.rgba_image(3 << 20), 6, {}),
        "table_p6_part_iters3": (_corpora.random_class(10, 2 << 20), 6, {"part_iters": 3}),
        "stored_pieces_p6": (lorem[:200000] + bytes(rng.integers(0, 256, size=300000, dtype=np.uint8)) + lorem[:150000]
                             + bytes(rng.integers(0, 256, size=100000, dtype=np.uint8)) + lorem[:100000], 6, {}),
    }


def encode(data, preset, over):
    import xz_amd
    opts = xz_amd.preset_options(preset)
    for k, v in over.items():
        setattr(opts, k, v)
    prm = o.params_for_gpu_options(opts)
    assert prm.enc_bits and prm.parser == 1 and prm.sa_window
    return o.orc_encode_block(data, prm)


def main():
    man = {}
    for name, (data, preset, over) in cases().items():
        raw = encode(data, preset, over)
        man[name] = {"in_size": len(data), "in_sha256": hashlib.sha256(data).hexdigest(),
                     "size": len(raw), "sha256": hashlib.sha256(raw).hexdigest()}
        print(name, len(data), "->", len(raw))
    with open(os.path.join(HERE, "own_definition.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
