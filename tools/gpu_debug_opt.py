#!/usr/bin/env python3
"""Debug helper: run optimal-parser cases of increasing size, each in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, xz_amd, _oracle as o
n, preset, parser, depth2, span = [int(x) for x in sys.argv[1:6]]
data = o.corpus_lorem(max(n, 1))[:n] if n < 300000 else o.corpus_mixed(n, 5)
opts = xz_amd.preset_options(preset, span_size=span)
opts.gpu_parser = parser
if depth2 >= 0:
    opts.gpu_sa_window = depth2
enc = xz_amd.Encoder()
prm = o.params_for_gpu_options(opts)
enc.trace_enable(n + 64)
t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
try:
    out, _ = enc.encode(t, opts=opts, block_size=1 << 20)
    got = out.cpu().numpy().tobytes()
except Exception as e:
    print("  encode raised:", e); got = b""
want = o.orc_xz_stream(data, prm, 1 << 20)
_, osym, _ = o.orc_encode_block(data, prm, want_trace=True)
gs, cnt = enc.trace_read(n + 64)
if len(gs):
    order = np.lexsort((gs[:, 1], gs[:, 0])); gs = gs[order][:, 1:]
if len(osym) and len(gs) and osym[0][0] == 0 and gs[0][0] != 0:
    osym = osym[1:]
fd = o.first_diff(got, want)
r, dec, nb = o.orc_xz_decode(got, n + 16) if got else (-999, b'', 0)
print('  roundtrip', r, dec == data)
print(f"  n={n} preset={preset} parser={parser} depth2={depth2}: out={len(got)} want={len(want)} {'IDENTICAL' if fd < 0 else 'DIFF@%%d' %% fd} symbols gpu={cnt} oracle={len(osym)}")
m = min(len(gs), len(osym))
neq = np.nonzero((gs[:m] != osym[:m]).any(axis=1))[0]
if len(neq):
    i = int(neq[0])
    for j in range(max(0, i - 3), min(m, i + 5)):
        print(f"    #{j} gpu={tuple(int(x) for x in gs[j])} oracle={tuple(int(x) for x in osym[j])}")
''' % (ROOT, ROOT)
cases = [(229001, 1, 1, -1), (400000, 2, 1, -1)]
W = 0xFFFFFFFF
for n, preset, parser, d2 in cases:
    print(f"case n={n} preset={preset}", flush=True)
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(preset), str(parser), str(d2), str(W)],
                           capture_output=True, text=True, timeout=300)
        print(r.stdout[-1500:], end="")
        if r.returncode != 0:
            print("  exit", r.returncode, r.stderr[-400:].replace("\n", " | "))
    except subprocess.TimeoutExpired:
        print("  TIMEOUT")
