#!/usr/bin/env python3
"""Per-span running times of the span kernel (a -DXZAMD_TIMING build leaves a record in every span's literal-coder
slice): distribution, and how it correlates with XCD / CU / start time.
usage: XZ_AMD_LIB=...timing.so tools/gpu_span_times.py [MiB] [span_cost]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1368
os.environ["XZAMD_NO_OVERLAP"] = "1"
import numpy as np, torch, xz_amd
n = mib << 20
t = torch.from_numpy(xz_amd.corpus_text(n, seed=1000)).cuda()
enc = xz_amd.Encoder(0)
opts = xz_amd.preset_options(6)
if len(sys.argv) > 2:
    opts.span_cost = int(sys.argv[2])
bs = xz_amd.mt_block_size(opts)
for _ in range(2):
    enc.encode(t, opts=opts)
st = enc.stats()
nb = (n + bs - 1) // bs
spb = bs // 65536 + 2
lit = enc.debug_fetch(8, nb * spb * 6144).reshape(nb * spb, 6144)
ok = lit[:, 0] == 0x54494D45
rec = lit[ok, :8].astype(np.int64)
ticks = rec[:, 1] + (rec[:, 2] << 32)
hw, xcc, nodes, start, length = rec[:, 3], rec[:, 4] & 15, rec[:, 5], rec[:, 6], rec[:, 7]
slot = np.nonzero(ok)[0]
print(f"spans {ok.sum()} span kernel {st.ms_encode - st.ms_find - st.ms_plan:.1f} ms; ticks avg {ticks.mean()/1e6:.1f}M min {ticks.min()/1e6:.1f}M max {ticks.max()/1e6:.1f}M std {ticks.std()/1e6:.1f}M")
per_node = ticks / np.maximum(nodes, 1)
print(f"nodes avg {nodes.mean():.0f} min {nodes.min()} max {nodes.max()}; ticks/node avg {per_node.mean():.1f} min {per_node.min():.1f} max {per_node.max():.1f}")
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1; simd = (hw >> 4) & 3
for name, key in (("xcc", xcc), ("se", se), ("cu", cu), ("simd", simd), ("block-pos k", slot % spb // 8), ("start>>", (start - start.min()) // max(1, (start.max() - start.min()) // 8 + 1))):
    ks = np.unique(key)
    print(name, " ".join(f"{k}:{per_node[key == k].mean():.0f}({(key == k).sum()})" for k in ks[:20]))
# waves sharing a CU (xcc, se, sh, cu) at the start: count by location
loc = xcc * 4096 + se * 512 + sh * 256 + cu * 16
u, c = np.unique(loc, return_counts=True)
print("spans per CU over the whole launch: min", c.min(), "max", c.max(), "CUs", len(u))
first = start < np.percentile(start, 100.0 * min(1.0, 4096 / len(start)))
print("first-round spans:", first.sum(), "ticks/node", per_node[first].mean(), "later:", per_node[~first].mean() if (~first).any() else None)
