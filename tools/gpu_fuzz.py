#!/usr/bin/env python3
"""Seeded option/data fuzz: HIP path vs the oracle (and the real reference where it applies), byte for byte."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, xz_amd, _oracle as o

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
enc = xz_amd.Encoder(0)
fails = 0
t0 = time.time()
for case in range(ncases):
    kind = int(rng.integers(0, 6))
    n = int(rng.integers(1, 600000)) if rng.random() < 0.7 else int(rng.integers(600000, 3500000))
    if kind == 0: data = o.corpus_lorem(n)
    elif kind == 1: data = o.corpus_mixed(n, int(rng.integers(1, 1000)))
    elif kind == 2: data = o.corpus_x86(n, int(rng.integers(1, 1000)))
    elif kind == 3: data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    elif kind == 4: data = (bytes(rng.integers(0, 4, 97, dtype=np.uint8)) * (n // 97 + 1))[:n]
    else: data = xz_amd.corpus_text(max(n, 4096), seed=int(rng.integers(1, 99))).tobytes()[:n]
    preset = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 9, 6 | 0x80000000]))
    opts = xz_amd.preset_options(preset)
    span = int(rng.choice([0, 0, 4096, 8192, 40000, 65536, 0xFFFFFFFF]))
    opts.span_size = span
    if rng.random() < 0.3 and not opts.gpu_sa_window:
        opts.gpu_parser = int(rng.integers(0, 2))       # exact finder with either parser
    if rng.random() < 0.2 and opts.gpu_sa_window:
        opts.gpu_sa_window = int(rng.integers(1, 6))
    if opts.gpu_sa_window and span == 0 and rng.random() < 0.6:
        # two-phase plan under stress: small parse pieces, small (or no) encode spans
        opts.span_cost = int(rng.choice([20000, 50000, 131072]))
        opts.span_bits = int(rng.choice([0, 100000]))
        opts.enc_span_bits = int(rng.choice([0, 100000, 400000, 1600000]))
        opts.part_iters = int(rng.choice([0, 0, 2, 3]))         # (round 6: partial parse iterations in front of the full one)
    if rng.random() < 0.2:
        opts.gpu_nice_len = int(rng.integers(max(4, opts.gpu_mf & 15), 274))
    if rng.random() < 0.15:
        opts.dict_size = int(rng.choice([4096, 65536, 1 << 20, 3 << 20]))
    if rng.random() < 0.15 and not opts.gpu_parser:
        lc = int(rng.integers(0, 5)); opts.lc = lc; opts.lp = int(rng.integers(0, 5 - lc)); opts.pb = int(rng.integers(0, 5))
    bcj = rng.random() < 0.2
    if bcj: opts.bcj = xz_amd.BCJ_X86
    bs = int(rng.choice([1 << 20, 200000, 65537, 1 << 16, 3 << 20]))
    if opts.span_size not in (0, 0xFFFFFFFF) and opts.span_size > bs: opts.span_size = 4096
    check = int(rng.choice([0, 1, 4]))
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    enc.trace_enable(n + 64)
    try:
        out, _ = enc.encode(t, opts=opts, block_size=bs, check=check)
    except Exception as e:  # noqa: BLE001
        print(f"case {case}: ENCODE ERROR {e} preset={preset:#x} span={span:#x} n={n}"); fails += 1; continue
    got = out.cpu().numpy().tobytes()
    ok = True
    msg = ""
    rr, dec = o.ref_decode(got, len(data) + 16)
    if not (rr == 1 and dec == data):
        ok = False; msg += " ROUNDTRIP"
    if not bcj:
        prm = o.params_for_gpu_options(opts)
        want = o.orc_xz_stream(data, prm, bs, check=check) if "check" in o.orc_xz_stream.__code__.co_varnames else None
        if want is not None and o.first_diff(got, want) != -1:
            ok = False; msg += f" ORACLE@{o.first_diff(got, want)}"
    gs, gcnt = enc.trace_read(n + 64)
    if not ok and not bcj and len(gs):
        # first differing LZMA symbol, Block by Block (device trace: span, offset in Block, back, len)
        prm = o.params_for_gpu_options(opts)
        sp = prm.span_size if prm.span_size else bs
        spb = (bs + sp - 1) // sp
        order = np.lexsort((gs[:, 1], gs[:, 0])); gs = gs[order]
        for b0 in range(0, n, bs):
            blk = data[b0:b0 + bs]
            _, osym, _ = o.orc_encode_block(blk, prm, want_trace=True)
            b = b0 // bs
            g = gs[(gs[:, 0] >= b * spb) & (gs[:, 0] < (b + 1) * spb)][:, 1:]
            if len(osym) and len(g) and osym[0][0] == 0 and g[0][0] != 0:
                osym = osym[1:]
            k = min(len(g), len(osym))
            dd = np.nonzero((g[:k] != osym[:k]).any(axis=1))[0]
            if len(dd) or len(g) != len(osym):
                i = int(dd[0]) if len(dd) else k
                msg += (f" | block {b}: symbol #{i} gpu {g[i].tolist() if i < len(g) else None} oracle {osym[i].tolist() if i < len(osym) else None}"
                        f" prev {g[max(i - 3, 0):i].tolist()} ctx {blk[int(osym[min(i, len(osym) - 1)][0]):int(osym[min(i, len(osym) - 1)][0]) + 16]!r}")
                break
    print(f"case {case}: {'ok ' if ok else 'FAIL' + msg} kind={kind} n={n} preset={preset:#x} span={span:#x} parser={opts.gpu_parser} d={opts.gpu_depth}/{opts.gpu_sa_window} "
          f"nice={opts.gpu_nice_len} dict={opts.dict_size} lc/lp/pb={opts.lc}/{opts.lp}/{opts.pb} bcj={int(bcj)} bs={bs} check={check} out={len(got)}", flush=True)
    fails += 0 if ok else 1
print(f"FUZZ seed {seed}: {ncases - fails} / {ncases} ok in {time.time()-t0:.0f} s")
sys.exit(1 if fails else 0)
