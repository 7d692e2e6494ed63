#!/usr/bin/env python3
"""Large-dictionary presets at real Block sizes: encode, decode with the REAL reference decoder, compare."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, xz_amd, _oracle as o
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
host = xz_amd.corpus_text(mib << 20, seed=77)
t = torch.from_numpy(host).cuda()
enc = xz_amd.Encoder()
for preset in [int(a, 0) for a in sys.argv[2:]] or [9, 7, 3]:
    opts = xz_amd.preset_options(preset)
    torch.cuda.synchronize(); t0 = time.time()
    out, binfo = enc.encode(t, opts=opts)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = enc.stats()
    data = out.cpu().numpy().tobytes()
    t1 = time.time()
    rr, dec = o.ref_decode(data, len(host) + 16)
    ok = rr == 1 and dec == host.tobytes()
    print(f"preset {preset:#x} {mib} MiB: {dt*1e3:.0f} ms ({mib*1.048576/dt:.0f} MB/s incl. first-call allocs), ratio {len(data)/len(host):.4f}, "
          f"blocks {len(binfo)}, dict {opts.dict_size>>20} MiB, chains {st.ms_chains:.0f} find {st.ms_find:.0f} encode {st.ms_encode:.0f} ms, "
          f"reference decoder round trip: {'OK' if ok else 'FAILED'} ({time.time()-t1:.1f} s)", flush=True)
    assert ok
