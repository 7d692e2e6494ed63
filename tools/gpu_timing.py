#!/usr/bin/env python3
"""Per-phase cycle breakdown of the span kernel: run with XZ_AMD_LIB=xz_amd/libxz_amd_timing.so XZAMD_TIMING=1
(a -DXZAMD_TIMING build).  usage: tools/gpu_timing.py [MiB=512] [preset=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, xz_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
preset = int(sys.argv[2], 0) if len(sys.argv) > 2 else 6
n = mib << 20
host = xz_amd.corpus_text(n, seed=1000)
t = torch.from_numpy(host).cuda()
enc = xz_amd.Encoder(0)
opts = xz_amd.preset_options(preset)
opts.span_size = xz_amd.SPAN_AUTO
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out, _ = enc.encode(t, opts=opts)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = enc.stats()
    print(f"[{it}] {n/dt/1e6:.1f} MB/s ratio {out.numel()/n:.4f} chains {st.ms_chains:.1f} find {st.ms_find:.1f} span {st.ms_encode-st.ms_find:.1f} total {st.ms_total:.1f} ms", flush=True)
