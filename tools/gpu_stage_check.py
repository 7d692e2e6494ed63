#!/usr/bin/env python3
"""Stage-by-stage device-vs-oracle check of the presets 4-9 path (run on the GPU box):
suffix order -> match-list records -> symbol stream -> bytes.  Prints the first divergence of every
stage in detail instead of stopping at the first failing assert."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import xz_amd  # noqa: E402
import _oracle as o  # noqa: E402


def check(enc, name, data, preset, bs, span, window=None):
    data = bytes(data)
    n = len(data)
    opts = xz_amd.preset_options(preset, span_size=span)
    if window is not None:
        opts.gpu_sa_window = window
    prm = o.params_for_gpu_options(opts)
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    enc.trace_enable(n + 64)
    try:
        out, _ = enc.encode(t, opts=opts, block_size=bs)
        got = out.cpu().numpy().tobytes()
        err = ""
    except Exception as e:  # noqa: BLE001
        got, err = b"", str(e)
    ok = True
    msg = []
    sa = enc.debug_fetch(1, n)
    rk = enc.debug_fetch(2, n)
    packed = opts.dict_size <= (1 << 23)
    lists = enc.debug_fetch(3, 8 * n).reshape(n, 8)
    lens = None if packed else enc.debug_fetch(4, 8 * n, "uint16").reshape(n, 8)
    for b0 in range(0, n, bs):
        blk = data[b0:b0 + bs]
        m = len(blk)
        osa, ork = o.orc_sa_dump(blk, opts.gpu_sa_depth)
        d = np.nonzero(sa[b0:b0 + m] != osa + b0)[0]
        if len(d):
            ok = False
            i = int(d[0])
            msg.append(f"SA block@{b0}: {len(d)} slots differ, first slot {i}: gpu pos {int(sa[b0 + i]) - b0} oracle {int(osa[i])}; "
                       f"gpu ctx {blk[int(sa[b0+i])-b0:int(sa[b0+i])-b0+12]!r} oracle ctx {blk[int(osa[i]):int(osa[i])+12]!r}")
            break
        if not (rk[b0:b0 + m] == ork + b0).all():
            ok = False
            msg.append(f"RANK block@{b0} differs")
            break
        want = o.orc_list_dump(blk, prm)
        g = lists[b0:b0 + m].astype(np.uint32).copy()
        if not packed:
            g[:, :7] = (lens[b0:b0 + m, :7].astype(np.uint32) << 23) | (g[:, :7] & 0x7FFFFF)   # compare in the oracle's packed form
            want = want.copy()
            want[:, :7] = (want[:, :7] & 0xFF800000) | (want[:, :7] & 0x7FFFFF)
        cnt = g[:, 7] & 0xFF
        for k in range(7):
            g[cnt <= k, k] = 0
        bad = np.nonzero((g != want).any(axis=1))[0]
        bad = bad[bad > 0]
        if len(bad):
            ok = False
            x = int(bad[0])
            def fmt(r):
                c = int(r[7]) & 0xFF
                return "cnt=%d l2=(%d,%d) " % (c, (int(r[7]) >> 8) & 0xFF, (int(r[7]) >> 16) & 0xFF) + \
                    " ".join("(%d,%d)" % (int(v) >> 23, int(v) & 0x7FFFFF) for v in r[:c])
            msg.append(f"LISTS block@{b0}: {len(bad)} positions differ, first x={x}: gpu {fmt(g[x])} | oracle {fmt(want[x])}")
            break
    want_s = o.orc_xz_stream(data, prm, bs)
    fd = o.first_diff(got, want_s)
    if fd >= 0 or err:
        ok = False
        msg.append(f"STREAM diff at byte {fd} (gpu {len(got)} B, oracle {len(want_s)} B) {err}")
        gs, cnt = enc.trace_read(n + 64)
        if len(gs) and bs >= n:
            _, osym, _ = o.orc_encode_block(data, prm, want_trace=True)
            order = np.lexsort((gs[:, 1], gs[:, 0]))
            gsym = gs[order][:, 1:]
            if len(osym) and len(gsym) and osym[0][0] == 0 and gsym[0][0] != 0:
                osym = osym[1:]
            k = min(len(gsym), len(osym))
            dd = np.nonzero((gsym[:k] != osym[:k]).any(axis=1))[0]
            if len(dd):
                i = int(dd[0])
                msg.append(f"  first differing symbol #{i}: gpu {gsym[i].tolist()} oracle {osym[i].tolist()} "
                           f"(prev gpu {gsym[max(i-2,0):i].tolist()})")
            else:
                msg.append(f"  symbols equal over {k} (gpu {len(gsym)} oracle {len(osym)})")
    else:
        enc.trace_read(16)
    print(f"[{name}] preset={preset:#x} n={n} bs={bs} span={span:#x} win={opts.gpu_sa_window}: {'OK' if ok else 'FAIL'}", flush=True)
    for m_ in msg:
        print("    " + m_, flush=True)
    return ok


def main():
    enc = xz_amd.Encoder()
    W = xz_amd.SPAN_WHOLE_BLOCK
    lorem = o.corpus_lorem(229001)
    mixed = o.corpus_mixed(300000, 5)
    text = xz_amd.corpus_text(400000, seed=7).tobytes()
    res = []
    res.append(check(enc, "tiny", lorem[:5000], 6, 1 << 20, W))
    res.append(check(enc, "lorem-whole", lorem, 6, 1 << 20, W))
    res.append(check(enc, "lorem-w2", lorem, 6, 1 << 20, W, window=2))
    res.append(check(enc, "mixed-whole", mixed, 6, 1 << 20, W))
    res.append(check(enc, "text-whole", text, 6, 1 << 20, W))
    res.append(check(enc, "text-span", text, 6, 1 << 20, 0))
    res.append(check(enc, "mixed-blocks", mixed, 6, 65536, 8192))
    res.append(check(enc, "mixed-ragged", mixed, 6, 100000, W))
    res.append(check(enc, "p4", mixed, 4, 1 << 20, 0))
    res.append(check(enc, "p7-unpacked", mixed, 7, 1 << 20, 0))
    res.append(check(enc, "p9e", text, 9 | xz_amd.PRESET_EXTREME, 1 << 20, 0))
    print("STAGE SUMMARY:", sum(res), "/", len(res), flush=True)
    # timing sample
    n = 512 << 20
    host = xz_amd.corpus_text(n, seed=1)
    t = torch.from_numpy(host).cuda()
    opts = xz_amd.preset_options(6)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        out, _ = enc.encode(t, opts=opts)
        torch.cuda.synchronize()
        dt = time.time() - t0
        st = enc.stats()
        print(f"[perf] preset 6, 512 MiB: {n/dt/1e6:.1f} MB/s ratio {out.numel()/n:.4f} | chains {st.ms_chains:.1f} find {st.ms_find:.1f} "
              f"encode {st.ms_encode:.1f} crc {st.ms_crc:.1f} total {st.ms_total:.1f} ms", flush=True)
    got = out.cpu().numpy().tobytes()
    r, dec = o.ref_decode(got, n + 16) if o.have_ref() else (1, host.tobytes())
    print("[perf] round trip through liblzma:", "OK" if (r == 1 and dec == host.tobytes()) else "FAIL", flush=True)
    return 0 if all(res) else 1


if __name__ == "__main__":
    sys.exit(main())
