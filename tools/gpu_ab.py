#!/usr/bin/env python3
"""A/B timing of whole encodes under different environment knobs in one process (one corpus, one torch import).
usage: tools/gpu_ab.py MiB PRESET [corpus=text|tar] name:ENV=V,ENV=V ...   (XZ_AMD_LIB selects the library build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, xz_amd
mib = int(sys.argv[1]); preset = int(sys.argv[2], 0)
args = sys.argv[3:]
corpus = "text"
if args and args[0].startswith("corpus="):
    corpus = args.pop(0).split("=", 1)[1]
n = mib << 20
if corpus == "elf":
    import bench
    host = bench.corpus_elf(n, 0)
else:
    host = xz_amd.corpus_tar(n, seed=1000) if corpus == "tar" else xz_amd.corpus_text(n, seed=1000)
t = torch.from_numpy(host).cuda()
KNOBS = ("XZAMD_PREFETCH_AFTER", "XZAMD_NO_OVERLAP", "XZAMD_SPAN_WAVES_PER_CU", "XZAMD_BATCH_MIB", "XZAMD_SA_COMPACT",
         "XZAMD_SPAN_COST", "XZAMD_SPAN_BITS", "XZAMD_SA_DEPTH", "XZAMD_BCJ", "XZAMD_WMAX_STD", "XZAMD_NICE")
for cfg in args or ["default:"]:
    name, _, envs = cfg.partition(":")
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    enc = xz_amd.Encoder(0)
    opts = xz_amd.preset_options(preset)
    if os.environ.get("XZAMD_SPAN_COST"):
        opts.span_cost = int(os.environ["XZAMD_SPAN_COST"])
    if os.environ.get("XZAMD_SA_DEPTH"):
        opts.gpu_sa_depth = int(os.environ["XZAMD_SA_DEPTH"])
    if os.environ.get("XZAMD_SPAN_BITS"):
        opts.span_bits = int(os.environ["XZAMD_SPAN_BITS"])
    if os.environ.get("XZAMD_BCJ"):
        opts.bcj = xz_amd.BCJ_X86
    if os.environ.get("XZAMD_NICE"):
        opts.gpu_nice_len = int(os.environ["XZAMD_NICE"])
    for it in range(int(os.environ.get("AB_REPS", "2"))):
        torch.cuda.synchronize(); t0 = time.time()
        out, _ = enc.encode(t, opts=opts)
        torch.cuda.synchronize(); dt = time.time() - t0
        st = enc.stats()
    import hashlib
    sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12] if os.environ.get("AB_SHA") else "-"
    print(f"{name:24s} {n/dt/1e6:8.1f} MB/s ratio {out.numel()/n:.5f} sha {sha} | chains {st.ms_chains:7.1f} find {st.ms_find:6.1f} (+{st.ms_find_overlapped:6.1f} lo) "
          f"plan {st.ms_plan:5.1f} span {st.ms_encode-st.ms_find-st.ms_plan:7.1f} crc {st.ms_crc:5.1f} total {st.ms_total:7.1f} ms | spans {st.spans} cost {st.span_cost_used} slots {st.wave_slots}", flush=True)
    enc.close()
    del enc
    torch.cuda.empty_cache()
