#!/usr/bin/env python3
"""Bring-up probe for the GPU box: runs the HIP path on a ladder of inputs, compares with the
oracle / real reference byte for byte and prints enough diagnostics (first differing byte,
first differing LZMA symbol) to debug a mismatch from one gpurun call.

    gpurun --timeout 600 -- 'python tools/gpu_probe.py > gpurun_out/probe.log 2>&1'
"""
import os
import sys
import time
import ctypes as C

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import xz_amd  # noqa: E402
import _oracle as o  # noqa: E402


def run_case(enc, name, data, preset, block_size, span, trace_on_fail=True, parser=None, depth2=None):
    data = bytes(data)
    opts = xz_amd.preset_options(preset, span_size=span)
    if parser is not None:
        opts.gpu_parser = parser
    if depth2 is not None:
        opts.gpu_sa_window = depth2
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda() if data else torch.empty(0, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    out, binfo = enc.encode(t, opts=opts, block_size=block_size)
    torch.cuda.synchronize()
    dt = time.time() - t0
    got = out.cpu().numpy().tobytes()
    st = enc.stats()
    prm = o.params_for_gpu_options(opts)
    want = o.orc_xz_stream(data, prm, block_size)
    fd = o.first_diff(got, want)
    r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    rt = (r == 0 and dec == data)
    line = (f"[{name}] preset={preset} n={len(data)} bs={block_size} span={span:#x} out={len(got)} want={len(want)} "
            f"{'IDENTICAL' if fd < 0 else 'DIFF@%d' % fd} roundtrip={'OK' if rt else 'FAIL(%d)' % r} "
            f"blocks={nb} wall={dt*1e3:.1f}ms enc={st.ms_encode:.2f}ms chains={st.ms_chains:.2f}ms "
            f"crc={st.ms_crc:.2f}ms asm={st.ms_assemble:.2f}ms")
    print(line, flush=True)
    if span == xz_amd.SPAN_WHOLE_BLOCK and preset <= 3 and parser is None and depth2 is None and o.have_ref() and block_size >= 4096:
        refs = o.ref_encode_mt(data, preset, threads=2, block_size=block_size)
        fr = o.first_diff(got, refs)
        print(f"    vs REAL reference liblzma {o.ref().ref_version().decode()}: "
              f"{'IDENTICAL' if fr < 0 else 'DIFF@%d' % fr} (ref {len(refs)} B)", flush=True)
    ok = fd < 0 and rt
    if not ok and trace_on_fail and len(data) <= (8 << 20):
        # symbol-level diagnosis on the first Block only
        blk = data[:block_size]
        _, osym, _ = o.orc_encode_block(blk, prm, want_trace=True)
        enc.trace_enable(len(blk) + 64)
        t1 = torch.frombuffer(bytearray(blk), dtype=torch.uint8).cuda()
        try:
            enc.encode(t1, opts=opts, block_size=block_size)
        except Exception as e:  # noqa: BLE001
            print("    trace encode raised:", e)
        gs, cnt = enc.trace_read(len(blk) + 64)
        # order GPU symbols by (span, pos)
        if len(gs):
            order = np.lexsort((gs[:, 1], gs[:, 0]))
            gs = gs[order][:, 1:]
        print(f"    symbols: gpu={cnt} oracle={len(osym)}")
        m = min(len(gs), len(osym))
        neq = np.nonzero((gs[:m] != osym[:m]).any(axis=1))[0]
        if len(neq):
            i = int(neq[0])
            lo = max(0, i - 3)
            print(f"    first differing symbol #{i}:")
            for j in range(lo, min(m, i + 4)):
                print(f"      #{j} gpu={tuple(int(x) for x in gs[j])} oracle={tuple(int(x) for x in osym[j])}")
        elif len(gs) != len(osym):
            print("    symbol streams agree on the common prefix; lengths differ")
        else:
            print("    symbol streams IDENTICAL -> the difference is in range coding / framing")
            # locate the first differing byte inside the first block payload
            print("    got [..]:", got[max(0, fd - 8):fd + 16].hex())
            print("    want[..]:", want[max(0, fd - 8):fd + 16].hex())
    return ok


def main():
    print("torch", torch.__version__, "device", torch.cuda.get_device_name(0), flush=True)
    print("lib", xz_amd.lib().xzamd_version().decode(), "have_ref", o.have_ref(), flush=True)
    enc = xz_amd.Encoder()
    W = xz_amd.SPAN_WHOLE_BLOCK
    lorem = o.corpus_lorem(229001)
    results = []
    results.append(run_case(enc, "tiny1", b"a", 1, 1 << 20, W))
    results.append(run_case(enc, "tiny5", b"abcab", 1, 1 << 20, W))
    results.append(run_case(enc, "lorem-4k", lorem[:4096], 1, 1 << 20, W))
    results.append(run_case(enc, "lorem", lorem, 1, 1 << 20, W))
    if not all(results):
        print("EARLY STOP: basic cases failing", flush=True)
        return 1
    mixed = o.corpus_mixed(400000, 5)
    rng = np.random.default_rng(7)
    rnd = bytes(rng.integers(0, 256, size=200000, dtype=np.uint8))
    big = o.corpus_lorem(1 << 20)
    sandwich = big[:150000] + rnd[:120000] + big[:100000]
    # exact HC4 finder + optimal parser
    results.append(run_case(enc, "exact-opt-lorem", lorem, 1, 1 << 20, W, parser=1))
    results.append(run_case(enc, "exact-opt-mixed", mixed, 2, 1 << 20, 65536, parser=1))
    # preset 6 mapping: suffix-neighbourhood finder + optimal parser
    results.append(run_case(enc, "p6-lorem-whole", lorem, 6, 1 << 20, W))
    results.append(run_case(enc, "p6-lorem-span", lorem, 6, 1 << 20, 0))
    results.append(run_case(enc, "p6-mixed", mixed, 6, 1 << 20, 0))
    results.append(run_case(enc, "p6-sandwich", sandwich, 6, 200000, 0))
    results.append(run_case(enc, "p6-rnd", rnd, 6, 1 << 20, 0))
    results.append(run_case(enc, "p9e-mixed", mixed, 9 | xz_amd.PRESET_EXTREME, 1 << 20, 0))
    results.append(run_case(enc, "p4-mixed", mixed, 4, 1 << 20, 0))
    results.append(run_case(enc, "lorem-p3", lorem, 3, 1 << 20, W))
    print("SUMMARY:", sum(results), "/", len(results), "cases OK", flush=True)

    # throughput sample
    for preset, mib in ((1, 256), (6, 256)):
        n = mib << 20
        host = xz_amd.corpus_text(n, seed=1)
        t = torch.from_numpy(host).cuda()
        opts = xz_amd.preset_options(preset)
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            out, _ = enc.encode(t, opts=opts)
            torch.cuda.synchronize()
            dt = time.time() - t0
            st = enc.stats()
            print(f"[perf] preset={preset} {mib} MiB text: wall {dt*1e3:.1f} ms = {n/dt/1e6:.1f} MB/s ratio {out.numel()/n:.4f} "
                  f"| chains {st.ms_chains:.1f} encode {st.ms_encode:.1f} crc {st.ms_crc:.1f} asm {st.ms_assemble:.1f} total {st.ms_total:.1f} ms",
                  flush=True)
        got = out.cpu().numpy().tobytes()
        r, dec, nb = o.orc_xz_decode(got, n + 16)
        print(f"[perf] roundtrip preset={preset}: {'OK' if (r == 0 and dec == host.tobytes()) else 'FAIL(%d)' % r} blocks={nb}", flush=True)
    return 0 if all(results) else 1


if __name__ == "__main__":
    sys.exit(main())
