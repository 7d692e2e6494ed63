#!/usr/bin/env python3
"""(CPU) One Block of a corpus class through the oracle restatement of the product path next to the real liblzma
(oracle/_ref) at the same preset: estimated bits per byte of the span plan, parse pieces / encode spans, sizes.

    python tools/oracle_probe.py CLASS PRESET [MiB] [param=value ...]
    CLASS   text | lorem | tar | logs | json | sqlite | dpkg_tar | elf_metadata | a key of tests/_corpora.NUMERIC_CLASSES /
            KNOWN_OUTSIDE (f32sine, sparse, relocs, rec13, ...)
    PRESET  6, 0x80000009 (= 9e), ...
    MiB     Block length (default: 24 at presets < 9e, 16 at 9e -- the sizes of test_size_within_tolerance_of_reference)
    param   an OrcParams field of the product mapping to override: span_cost=196608 span_bits=800000 enc_bits=2400000 pb=4 ...
    XZAMD_ORACLE_DIR=/dir  load liboracle.so from there (a build of oracle/*.c with other constants, e.g. ORC_WARM)

How the round-5 figures "through the oracle" of DESIGN.md 3.4 / 4 / 7 were taken (cost of a piece start, the bit bound of a
piece on highly compressible Blocks, pb = 3 / 4, walk and pre-roll lengths).  Test infrastructure, not product code."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _corpora  # noqa: E402
import _oracle as o  # noqa: E402
import xz_amd  # noqa: E402


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    which, preset = sys.argv[1], int(sys.argv[2], 0)
    rest = sys.argv[3:]
    n = (16 << 20) if preset >> 31 else (24 << 20)
    if rest and "=" not in rest[0]:
        n = int(rest[0]) << 20
        rest = rest[1:]
    if os.environ.get("XZAMD_ORACLE_DIR"):
        o.ORACLE_DIR = os.environ["XZAMD_ORACLE_DIR"]          # (its _ref/ must exist there too: a symlink will do)
    gens = {"text": lambda k: xz_amd.corpus_text(k, seed=1000).tobytes(), "lorem": o.corpus_lorem,
            "tar": lambda k: xz_amd.corpus_tar(k).tobytes(), "logs": _corpora.logs, "json": _corpora.json_records,
            "sqlite": _corpora.sqlite_file, "dpkg_tar": _corpora.dpkg_tar, "elf_metadata": _corpora.elf_metadata}
    gens.update(_corpora.NUMERIC_CLASSES)
    gens.update(_corpora.REVIEW_CLASSES)
    gens.update({k: v[0] for k, v in _corpora.KNOWN_OUTSIDE.items()})
    if which.startswith("rnd"):
        gens[which] = lambda k: _corpora.random_class(int(which[3:]), k)     # draw number N of the random class generator
    data = gens[which](n)
    if data is None:
        raise SystemExit(f"{which}: not available on this image")
    opts = xz_amd.preset_options(preset)
    pb = opts.pb
    for a in rest:
        k, v = a.split("=")
        if k == "pb":
            pb = int(v, 0)
            opts.pb = pb
    prm = o.params_for_gpu_options(opts)
    for a in rest:
        k, v = a.split("=")
        if k != "pb":
            setattr(prm, k, int(v, 0))
    _, bits, _ = o.orc_span_plan(data, prm)
    starts, estarts = o.orc_piece_plan(data, prm) if prm.enc_bits else (o.orc_span_plan(data, prm)[2], [])
    t = time.time()
    raw = o.orc_encode_block(data, prm)
    dt = time.time() - t
    r, dec = o.ref_raw_decode(raw, prm.dict_size, len(data) + 16)
    assert r == 1 and dec == data, "round trip through the reference decoder failed"
    refp = o.OrcParams(opts.dict_size, opts.lc, opts.lp, pb, opts.nice_len, 0x14, 0, 0, 0, 0)
    ref = len(o.ref_raw_encode(data, refp, mode=2))
    print(f"{which} preset {preset:#x} {len(data) >> 20} MiB {' '.join(rest)}: est {bits[16:].sum() / max(1, len(data) - 65536):.2f} bits/byte, "
          f"{len(starts)} pieces, {len(estarts)} encode spans, ours {len(raw)} vs liblzma {ref}: {100 * (len(raw) / ref - 1):+.2f} %  ({dt:.0f} s)")


if __name__ == "__main__":
    main()
