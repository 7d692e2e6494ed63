#!/usr/bin/env python3
"""Per-span running times of the span kernel (a -DXZAMD_TIMING build leaves a record in every span's literal-coder
slice): distribution, and how it correlates with XCD / CU / start time.
usage: XZ_AMD_LIB=...timing.so tools/gpu_span_times.py [MiB] [span_cost] [preset] [corpus=text|elf|tar] [bcj]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1368
os.environ["XZAMD_NO_OVERLAP"] = "1"
import numpy as np, torch, xz_amd
n = mib << 20
preset = int(sys.argv[3], 0) if len(sys.argv) > 3 else 6
corpus = sys.argv[4] if len(sys.argv) > 4 else "text"
if corpus == "elf":
    import bench
    host = bench.corpus_elf(n, 0)
elif corpus == "tar":
    host = xz_amd.corpus_tar(n, seed=1000)
else:
    host = xz_amd.corpus_text(n, seed=1000)
t = torch.from_numpy(host).cuda()
enc = xz_amd.Encoder(0)
opts = xz_amd.preset_options(preset)
if len(sys.argv) > 2 and int(sys.argv[2]):
    opts.span_cost = int(sys.argv[2])
if len(sys.argv) > 5:
    opts.bcj = xz_amd.BCJ_X86
bs = xz_amd.mt_block_size(opts)
for _ in range(1):
    enc.encode(t, opts=opts)
st = enc.stats()
nb = (n + bs - 1) // bs
spb = bs // 65536 + 2
lit = enc.debug_fetch(8, nb * spb * 6144).reshape(nb * spb, 6144)
tab = enc.debug_fetch(5, 2 * nb * spb).reshape(nb * spb, 2)
est = enc.debug_fetch(7, 2 * nb * ((bs + 4095) // 4096)).reshape(2, nb, (bs + 4095) // 4096)
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

# the slowest spans: what the plan thought of them
order = np.argsort(-ticks)[:12]
cpb = (bs + 4095) // 4096
print("slowest spans: slot start len ticks(M) nodes syms? est_work est_bits ticks/node nodes/len")
for i in order:
    sl = slot[i]
    st0, en0 = int(tab[sl, 0]), int(tab[sl, 1])
    b = st0 // bs
    c0, c1 = (st0 - b * bs) // 4096, (en0 - b * bs + 4095) // 4096
    ew, eb = int(est[0, b, c0:c1].sum()), int(est[1, b, c0:c1].sum())
    print(sl, st0, en0 - st0, round(ticks[i] / 1e6), int(nodes[i]), ew, eb, round(per_node[i]), round(nodes[i] / max(1, en0 - st0), 3))
print("all spans: est work per span min/mean/max", end=" ")
ews = []
for i in range(len(slot)):
    sl = slot[i]; st0, en0 = int(tab[sl, 0]), int(tab[sl, 1]); b = st0 // bs
    ews.append(int(est[0, b, (st0 - b * bs) // 4096:(en0 - b * bs + 4095) // 4096].sum()))
ews = np.array(ews)
print(ews.min(), int(ews.mean()), ews.max(), "corr(est work, ticks) =", round(float(np.corrcoef(ews, ticks)[0, 1]), 3), "corr(nodes, ticks) =", round(float(np.corrcoef(nodes, ticks)[0, 1]), 3))
