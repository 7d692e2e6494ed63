#!/usr/bin/env python3
"""One encode of N MiB synthetic text at a preset (for rocprofv3 runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, xz_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
preset = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
host = xz_amd.corpus_text(mib << 20, seed=1000)
t = torch.from_numpy(host).cuda()
enc = xz_amd.Encoder()
opts = xz_amd.preset_options(preset)
if os.environ.get('XZAMD_SPAN_AUTO'):
    opts.span_size = xz_amd.SPAN_AUTO
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.time()
    out, _ = enc.encode(t, opts=opts)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = enc.stats()
    print(f"preset {preset} {mib} MiB: {dt*1e3:.1f} ms, {mib*1.048576/dt:.1f} MB/s, ratio {out.numel()/(mib<<20):.4f}, encode {st.ms_encode:.1f} ms (find {st.ms_find:.1f}) chains {st.ms_chains:.1f} ms", flush=True)
