#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LZMA2 Block encoder.

Metric (BASELINE.json): compress MB/s (10^6 uncompressed bytes per wall second) + ratio vs
`xz -T0 -6`, 4 GiB synthetic enwik-style input per GPU, preset -6 options (8 MiB dictionary,
24 MiB Blocks).  One "step" = one full pass of the hot path (match-finder build, span encode,
CRC64, assembly into a complete .xz Stream) over the batch already resident in HBM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N>1: weak scaling -- every rank encodes its own shard of Blocks on its own GPU (no data-path
collective), then the encoded Blocks are gathered to rank 0 over RCCL (send/recv of
variable-length byte tensors + the 16-byte Index records) and rank 0 frames the Stream.  The gather
is inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import xz_amd  # noqa: E402
from xz_amd import parallel  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def baseline_metric():
    """The metric string of BASELINE.json (kept verbatim so the line can be matched to it)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "compress MB/s + ratio vs xz -T0 -6, 4 GiB input, 1/2/4/8 MI355X"


def span_kernel_name(opts, pmc=False):
    """Name of the dominant kernel for these options (template args: finder source, parser)."""
    finder = 2 if opts.gpu_parser else 0
    sep = ", " if pmc else ","
    return "k_span_encode_t<%d%s%s>" % (finder, sep, "true" if opts.gpu_parser else "false")


def pmc_traffic(opts):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE in separate passes, tools/prof_bench.sh -> profiles/*pmc_summary.json).
    PMC counters cannot be collected from inside this process; None when no profile of the same
    kernel variant is present."""
    import glob
    want = span_kernel_name(opts, pmc=True)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        for k, e in d.items():
            if want in k and "hbm_bytes_per_dispatch_uncorrected" in e:
                return int(e["hbm_bytes_per_dispatch_uncorrected"]), os.path.basename(f)
    return None, None


def corpus_elf(n, rank):
    """x86-64 ELF shared objects found on the box, in sorted order, cycled with a per-cycle byte perturbation
    so cycles are not identical (SURVEY.md 8d C4/C5 recipe).  Not synthetic: bench.py says so in `data`."""
    import glob
    files = sorted(glob.glob("/opt/rocm/lib/*.so*") + glob.glob("/usr/lib/x86_64-linux-gnu/*.so*"))
    files = [f for f in files if os.path.isfile(f) and not os.path.islink(f) and os.path.getsize(f) > 65536]
    if not files:
        raise SystemExit("no ELF files found for --corpus elf")
    out = np.empty(n, dtype=np.uint8)
    pos, cycle, i = 0, 0, rank % len(files)
    while pos < n:
        a = np.fromfile(files[i], dtype=np.uint8, count=min(n - pos, 256 << 20))
        if cycle:
            a = a.copy()
            a[cycle::4099] ^= np.uint8(cycle & 0xFF)
        out[pos:pos + len(a)] = a
        pos += len(a)
        i += 1
        if i == len(files):
            i = 0
            cycle += 1
    return out


def cpu_baseline(sample, preset, bcj=False):
    """Reference liblzma (oracle/_ref, the real 5.8.3 sources) MT encoder on the host cores, timed on a
    bounded sample of the same workload.  Test infrastructure used as a reported baseline only."""
    try:
        import _oracle as o
        if not o.have_ref():
            return None, None
        cores = int(o.ref().ref_cputhreads())
        t0 = time.time()
        enc = (o.ref_encode_mt_x86 if bcj else o.ref_encode_mt)(sample, preset, threads=max(cores, 1), block_size=0)
        dt = time.time() - t0
        return {"value": round(len(sample) / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": "reference",
                "sample": f"first {len(sample) >> 20} MiB of rank 0's input, liblzma 5.8.3 lzma_stream_encoder_mt "
                          f"preset {preset & 31}{'e' if preset >> 31 else ''}{' + x86 BCJ' if bcj else ''} threads={cores} default block size, {dt:.1f} s wall",
                "ratio": round(len(enc) / max(len(sample), 1), 5)}, enc
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-mib", type=int, default=4096, help="input MiB per GPU")
    ap.add_argument("--preset", type=lambda v: int(v, 0), default=6, help="0-9, | 0x80000000 for -e")
    ap.add_argument("--span-kib", type=int, default=0, help="0 = library default")
    ap.add_argument("--block-mib", type=int, default=0, help="0 = lzma_mt_block_size of the preset (BASELINE configs[1]: 16)")
    ap.add_argument("--parser", choices=["default", "fast", "optimal"], default="default",
                    help="device parser override (default: what the preset maps to)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bcj", action="store_true", help="chain {x86 BCJ, LZMA2} (SURVEY.md 8d config C5)")
    ap.add_argument("--corpus", choices=["text", "elf"], default="text",
                    help="text: seeded synthetic enwik-style text (the headline workload); elf: the x86-64 shared "
                         "objects present on the box, concatenated and cycled (config C5's input)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    n = args.size_mib << 20
    opts = xz_amd.preset_options(args.preset)
    if args.span_kib:
        opts.span_size = args.span_kib << 10
    if args.parser != "default":
        opts.gpu_parser = 1 if args.parser == "optimal" else 0
    block_size = (args.block_mib << 20) if args.block_mib else xz_amd.mt_block_size(opts)

    if args.bcj:
        opts.bcj = xz_amd.BCJ_X86
    host = corpus_elf(n, rank) if args.corpus == "elf" else xz_amd.corpus_text(n, seed=1000 + rank)
    data = torch.from_numpy(host).to(dev)
    enc = xz_amd.Encoder(local_rank)
    out_buf = torch.empty(xz_amd.lib().xzamd_stream_buffer_bound(n, block_size) + 64, dtype=torch.uint8, device=dev)

    def step():
        if world == 1:
            out, binfo = enc.encode(data, opts=opts, block_size=block_size, out=out_buf)
            return out, binfo
        out, binfo = enc.encode(data, opts=opts, block_size=block_size, out=out_buf, blocks_only=True)
        stream = parallel.gather_stream(out, binfo, check=xz_amd.CHECK_CRC64)
        return stream, binfo

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc_ms = 0.0
    launches = 0
    out = None
    for _ in range(args.steps):
        out, binfo = step()
        st = enc.stats()
        enc_ms += st.ms_encode - st.ms_find      # the span kernel alone (k_find_t is timed separately)
        launches += st.encode_launches
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = enc.stats()
    total_in = n * world * args.steps
    value = total_in / elapsed / 1e6
    # roofline of the dominant kernel (k_span_encode): algorithmic bytes = uncompressed in +
    # compressed out per launch, divided by its HIP-event time on its own stream.
    alg_bytes = (st.in_bytes + st.out_bytes) * args.steps
    achieved = alg_bytes / (enc_ms / 1e3) / 1e9 if enc_ms > 0 else 0.0

    if rank == 0:
        local_out_bytes = int(st.out_bytes)
        res = {
            "metric": baseline_metric(),
            "value": round(value, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if args.corpus == "text" else "x86-64 ELF shared objects of the box, concatenated/cycled",
            "config": {
                "workload": f"preset -{args.preset & 31}{'e' if args.preset >> 31 else ''} options (dict {opts.dict_size >> 20} MiB, {block_size >> 20} MiB Blocks, "
                            f"CRC64{', x86 BCJ + LZMA2' if args.bcj else ''}), {args.size_mib} MiB "
                            f"{'synthetic enwik-style text' if args.corpus == 'text' else 'ELF shared objects'} per GPU, input resident in HBM, "
                            f"output = complete .xz Stream in HBM",
                "device_match_finder": ((f"suffix-neighbourhood finder (32-byte-prefix suffix order, {opts.gpu_sa_window} slots per side + hash2/3/4/8)"
                                         if opts.gpu_sa_window else f"HC{opts.gpu_mf & 15} depth {opts.gpu_depth} (sort-built chains)")
                                        + f", nice {opts.gpu_nice_len}"),
                "device_parser": ("windowed optimal parser (232-node DP, exact prices, compound edges) over per-position match lists" if opts.gpu_parser
                                  else "lzma_lzma_optimum_fast semantics (greedy + 1-byte lazy)"),
                "span_kib": (opts.span_size or (131072 if opts.gpu_parser else 65536)) >> 10,
                "parallelism": f"{world} x (one wavefront per span, {int(st.spans)} spans per GPU)",
            },
            "ratio": {"ours": round(local_out_bytes / n, 5)},
            "roofline": {
                "bound": "hbm",
                "kernel": span_kernel_name(opts),
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": pmc_traffic(opts)[0],
                "traffic_source": pmc_traffic(opts)[1],
                "algorithmic_bytes_per_launch": int(alg_bytes / max(launches, 1)),
                "avg_launch_ms": round(enc_ms / max(launches, 1), 3),
                "launches": launches,
            },
            "stage_ms_last_step": {"chains": round(st.ms_chains, 2), "find": round(st.ms_find, 2),
                                   "span_encode": round(st.ms_encode - st.ms_find, 2),
                                   "crc": round(st.ms_crc, 2), "assemble": round(st.ms_assemble, 2),
                                   "total": round(st.ms_total, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                import _oracle as o
                cores = int(o.ref().ref_cputhreads()) if o.have_ref() else 1
            except Exception:  # noqa: BLE001
                cores = 1
            sample_n = min(n, min(max(1, cores), 32) * block_size)
            cb, ref_enc = cpu_baseline(host[:sample_n], args.preset, args.bcj)
            if cb is not None:
                res["cpu_baseline"] = cb
                # our ratio on the same sample + bit-exact round trip of that output through the
                # REAL reference decoder (first 64 MiB only: the CPU decoder is the slow side)
                try:
                    import _oracle as o
                    s_out, _ = enc.encode(data[:sample_n], opts=opts, block_size=block_size)
                    res["ratio"]["ours_on_sample"] = round(s_out.numel() / sample_n, 5)
                    if cb.get("ratio"):
                        res["ratio"][f"reference_xz_T0_{args.preset & 31}{'e' if args.preset >> 31 else ''}_on_sample"] = cb["ratio"]
                        res["ratio"]["size_vs_reference_pct"] = round(100.0 * (s_out.numel() / sample_n / cb["ratio"] - 1), 2)
                    vn = min(sample_n, 64 << 20)
                    v_out, _ = enc.encode(data[:vn], opts=opts, block_size=block_size)
                    rr, dec = o.ref_decode(v_out.cpu().numpy().tobytes(), vn + 16)
                    res["roundtrip_reference_decoder"] = bool(rr == 1 and dec == host[:vn].tobytes())
                except Exception as e:  # noqa: BLE001
                    res["roundtrip_reference_decoder"] = f"failed: {e}"
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
