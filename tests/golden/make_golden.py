#!/usr/bin/env python3
"""Regenerates tests/golden/ (run in the dev container where /root/reference exists).

* copies the reference's own decoder fixture files that concern the LZMA2 /
  Block path (tests/files/README:48-105,281-) -- data, not source code -- into
  tests/golden/ref_files/;
* records, via the REAL reference library (oracle/_ref), what each must decode
  to (size + sha256) or that it must be rejected;
* records sha256 of the reference's raw LZMA2 encoder output for the reference
  test corpora (tests/create_compress_files.c) at presets 0-3, so the oracle
  encoder stays pinned even where oracle/_ref is unavailable.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as o  # noqa: E402

REF_FILES = "/root/reference/tests/files"
GOOD = ["good-0-empty.xz", "good-1-check-none.xz", "good-1-check-crc32.xz", "good-1-check-crc64.xz",
        "good-1-block_header-1.xz", "good-1-block_header-2.xz", "good-1-block_header-3.xz",
        "good-1-lzma2-1.xz", "good-1-lzma2-2.xz", "good-1-lzma2-3.xz", "good-1-lzma2-4.xz",
        "good-1-lzma2-5.xz", "good-2-lzma2.xz"]
BAD = ["bad-1-lzma2-%d.xz" % i for i in range(1, 12)] + ["bad-1-check-crc64.xz", "bad-1-block_header-4.xz",
       "bad-2-index-1.xz", "bad-1-stream_flags-2.xz"]


def main():
    assert o.have_ref(), "needs oracle/_ref (the real reference build)"
    os.makedirs(os.path.join(HERE, "ref_files"), exist_ok=True)
    man = {"reference_version": o.ref().ref_version().decode(), "decode": {}, "encode": {}}
    for name in GOOD + BAD:
        src = os.path.join(REF_FILES, name)
        shutil.copyfile(src, os.path.join(HERE, "ref_files", name))
        blob = open(src, "rb").read()
        r, dec = o.ref_decode(blob, 1 << 20)
        if name in GOOD:
            assert r == 1, (name, r)
            man["decode"][name] = {"ok": True, "size": len(dec), "sha256": hashlib.sha256(dec).hexdigest()}
        else:
            assert r != 1, (name, r)
            man["decode"][name] = {"ok": False}
    corpora = {"text": o.corpus_lorem(229001), "abc": o.corpus_abc(), "random": o.corpus_random()}
    for cname, data in corpora.items():
        man["encode"][cname] = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest(), "raw_lzma2": {}}
        for preset in (0, 1, 2, 3):
            prm, _ = o.params_for_preset(preset)
            enc = o.ref_raw_encode(data, prm, mode=1)
            man["encode"][cname]["raw_lzma2"][str(preset)] = {
                "size": len(enc), "sha256": hashlib.sha256(enc).hexdigest()}
        mt = o.ref_encode_mt(data, 1, threads=2, block_size=65536)
        man["encode"][cname]["mt_preset1_bs64k"] = {"size": len(mt), "sha256": hashlib.sha256(mt).hexdigest()}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print("wrote", len(GOOD) + len(BAD), "fixtures and manifest.json")


if __name__ == "__main__":
    main()
