"""Bring-up check of the two-phase path on the GPU: plan, recorded symbols and bytes vs the oracle; timing of the stages."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as o
import xz_amd


def check(enc, data, preset, block_size, label):
    opts = xz_amd.preset_options(preset)
    t = torch.from_numpy(data).to("cuda:0")
    out, _ = enc.encode(t, opts=opts, block_size=block_size)
    got = out.cpu().numpy().tobytes()
    st = enc.stats()
    n = len(data)
    nb = (n + block_size - 1) // block_size
    prm = o.params_for_gpu_options(opts)
    raw = data.tobytes()
    r, dec, _ = o.orc_xz_decode(got, n + 16)
    ok_rt = r == 0 and dec == raw
    # plan + symbols of the last block of the (single) batch are in the debug buffers; compare block 0 when nb == 1
    msg = f"{label}: n={n} blocks={nb} out={len(got)} roundtrip={ok_rt} pieces={st.spans} enc_spans={st.enc_spans} " \
          f"ms seed {st.ms_seed:.1f} parse {st.ms_parse:.1f} code {st.ms_code:.1f} total {st.ms_total:.1f}"
    print(msg, flush=True)
    if not ok_rt:
        print("  decode rc", r)
        return False
    if n <= (8 << 20):
        sl = enc.debug_fetch(9, n, "uint16")
        sd = enc.debug_fetch(10, n, "uint32")
        ok = True
        for b in range(nb):
            blk = data[b * block_size:(b + 1) * block_size]
            osl, osd = o.orc_parse_dump(blk, prm)
            gsl = sl[b * block_size:b * block_size + len(blk)]
            gsd = sd[b * block_size:b * block_size + len(blk)]
            # walk the oracle's symbols
            p = 0
            bad = None
            while p < len(blk):
                if osl[p] != gsl[p] or osd[p] != gsd[p]:
                    bad = p
                    break
                p += max(1, int(osl[p]) & 0x7FFF)
            if bad is not None:
                print(f"  block {b}: symbol records differ at {bad}: oracle ({osl[bad]}, {osd[bad]:#x}) gpu ({gsl[bad]}, {gsd[bad]:#x})")
                ok = False
                break
        print("  symbol records identical:", ok)
        want = o.orc_xz_stream(raw, prm, block_size)
        same = want == got
        print("  stream identical to oracle:", same, "" if same else f"(first diff {o.first_diff(got, want)}, sizes {len(got)} {len(want)})")
        return ok and same
    return True


def main():
    enc = xz_amd.Encoder(0)
    ok = True
    ok &= check(enc, xz_amd.corpus_text(3 << 20), 6, 1 << 20, "text 3 MiB / 1 MiB blocks p6")
    ok &= check(enc, xz_amd.corpus_lorem(2 << 20), 6, 2 << 20, "lorem 2 MiB p6")
    ok &= check(enc, xz_amd.corpus_tar(4 << 20), 6, 4 << 20, "tar 4 MiB p6")
    ok &= check(enc, xz_amd.corpus_text(1 << 20), 9 | xz_amd.PRESET_EXTREME, 1 << 20, "text 1 MiB p9e")
    rnd = np.random.default_rng(1).integers(0, 256, 1 << 20, dtype=np.uint8)
    ok &= check(enc, rnd, 6, 1 << 20, "random 1 MiB p6")
    ok &= check(enc, np.zeros(3 << 20, dtype=np.uint8), 6, 2 << 20, "zeros 3 MiB p6")
    ok &= check(enc, xz_amd.corpus_text(100), 6, 1 << 20, "text 100 B")
    big = xz_amd.corpus_text(1368 << 20)
    t0 = time.time()
    ok &= check(enc, big, 6, 24 << 20, "text 1368 MiB p6")
    print("wall", time.time() - t0)
    print("ALL OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
