#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LZMA2 Block encoder.

Metric (BASELINE.json): compress MB/s (10^6 uncompressed bytes per wall second) + ratio vs
`xz -T0 -6`, 4 GiB synthetic enwik-style input per GPU, preset -6 options (8 MiB dictionary,
24 MiB Blocks).  One "step" = one full pass of the hot path (match-finder build, span encode,
CRC64, assembly into a complete .xz Stream) over the batch already resident in HBM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 spawns its own N ranks (torch.distributed.run, one process per
GPU).  Every rank encodes its own shard of Blocks on its own GPU (no data-path collective), then the
encoded Blocks are gathered to rank 0 over RCCL (send/recv of variable-length byte tensors + the 16-byte
Index records) and rank 0 frames the Stream.  The gather is inside the timed region.
--scaling strong (default): --size-mib in total, whole Blocks dealt to the ranks in order (the metric's 4 GiB
at 1/2/4/8 GPUs); --scaling weak: --size-mib per GPU.  Prints ONE JSON line on rank 0, with `roofline`,
`cpu_baseline` (reference liblzma -T0 on the host cores, bounded wall time) and `host_to_host` (the same
job through lzma_code on host buffers, PCIe inclusive).
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


def two_phase(opts):
    """Parse pieces + encode spans (the default of the optimal-parser presets) instead of the single-phase span kernel."""
    return bool(opts.gpu_parser and opts.gpu_sa_window and opts.span_cost and opts.enc_span_bits
                and opts.span_size in (xz_amd.SPAN_DEFAULT, xz_amd.SPAN_AUTO))


def span_kernel_name(opts, pmc=False):
    """Name of the dominant kernel for these options (template args: finder source, parser, parser window)."""
    wm = 384            # one parser window for every option set since round 5 (lzma_kernels.hip: WMAX_STD == WMAX_LONG)
    if two_phase(opts):
        return "k_parse_pieces<%du" % wm if pmc else "k_parse_pieces<%d>" % wm       # (rocprofv3: `<384u, true>` = packed list records)
    finder = 2 if opts.gpu_parser else 0
    sep = ", " if pmc else ","
    if pmc:          # as rocprofv3 prints it
        return "k_span_encode_t<%d%s%s%s%du>" % (finder, sep, "true" if opts.gpu_parser else "false", sep, wm)
    return "k_span_encode_t<%d,%s,%d>" % (finder, "true" if opts.gpu_parser else "false", wm)


def pmc_traffic(opts, corpus, preset, bcj):
    """HBM bytes per BIG launch (a full timed batch) of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE in separate passes, tools/prof_bench.sh -> tools/pmc_summary.py ->
    profiles/*pmc_summary.json).  PMC counters cannot be collected from inside this process.  A figure is quoted only
    from a profile of the SAME kernel variant, corpus, preset and filter chain (`_meta` of the summary); else None."""
    import glob
    want = span_kernel_name(opts, pmc=True)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        meta = d.get("_meta") or {}
        if not meta or meta.get("corpus") != corpus or int(str(meta.get("preset", "6")), 0) != preset or bool(meta.get("bcj")) != bool(bcj):
            continue
        for k, e in d.items():
            if want in k and isinstance(e, dict) and "hbm_bytes_per_big_launch_uncorrected" in e:
                return int(e["hbm_bytes_per_big_launch_uncorrected"]), os.path.basename(f)
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


def cpu_limits():
    """What the host offers: CPUs in the affinity mask, the cgroup CPU quota (cpu.max), hardware threads."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["sched_getaffinity"] = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        info["sched_getaffinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q < 0 else q / per
            info["cgroup_cpu_max"] = " ".join(txt)
            break
        except Exception:  # noqa: BLE001
            continue
    info["cgroup_quota_cpus"] = quota
    return info


def cpu_baseline(host, preset, bcj, block_size, seconds=20.0):
    """Reference liblzma (oracle/_ref, the real 5.8.3 sources) on the host cores, beside the GPU number.
    `xz -T0`-equivalent: lzma_stream_encoder_mt with threads = lzma_cputhreads() over the WHOLE input of
    rank 0 (every worker has a Block to chew on), timed for a bounded wall time: value = input bytes the
    workers processed (lzma_get_progress) / wall.  Plus the -T1 per-core figure and the host's CPU limits.
    Test infrastructure used as a reported baseline only."""
    import ctypes as C
    try:
        import _oracle as o
        if not o.have_ref():
            return None
        f = o.ref().ref_encode_mt_timed
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_double,
                      C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        pin, el, th = C.c_uint64(), C.c_double(), C.c_uint32()
        lim = cpu_limits()
        r = f(host.ctypes.data, host.size, preset, 1 if bcj else 0, 0, 0, seconds, C.byref(pin), C.byref(el), C.byref(th))
        if r != 1:
            return {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": f"lzma_code failed: {r}"}
        nblocks = (host.size + block_size - 1) // block_size
        busy = min(int(th.value), int(nblocks))
        # what the workers can actually run on: the affinity mask and the cgroup CPU quota bound the cores in use
        usable = min(x for x in (busy, lim.get("sched_getaffinity") or busy, int(lim["cgroup_quota_cpus"] + 0.5) if lim.get("cgroup_quota_cpus") else busy))
        mt = pin.value / el.value / 1e6
        pin1, el1, th1 = C.c_uint64(), C.c_double(), C.c_uint32()
        f(host.ctypes.data, host.size, preset, 1 if bcj else 0, 1, 0, min(seconds, 6.0), C.byref(pin1), C.byref(el1), C.byref(th1))
        t1 = pin1.value / max(el1.value, 1e-9) / 1e6
        # the same encoder with threads = the CPUs this process may use (what `xz -T<usable>` would run): no oversubscription
        pinu, elu, thu = C.c_uint64(), C.c_double(), C.c_uint32()
        f(host.ctypes.data, host.size, preset, 1 if bcj else 0, max(1, usable), 0, min(seconds, 10.0), C.byref(pinu), C.byref(elu), C.byref(thu))
        tu = pinu.value / max(elu.value, 1e-9) / 1e6
        return {"value": round(mt, 2), "unit": "MB/s", "cores": usable, "kind": "reference",
                "sample": (f"liblzma 5.8.3 lzma_stream_encoder_mt preset {preset & 31}{'e' if preset >> 31 else ''}{' + x86 BCJ' if bcj else ''}, "
                           f"threads={int(th.value)} (lzma_cputhreads, as xz -T0), whole {host.size >> 20} MiB input of rank 0 on offer = {nblocks} Blocks "
                           f"-> {busy} worker threads with a Block each on {usable} usable CPUs (affinity mask / cgroup quota), "
                           f"timed {el.value:.1f} s wall, {pin.value >> 20} MiB processed (lzma_get_progress)"),
                "worker_threads": busy,
                "threads_equal_usable_cores": {"value": round(tu, 2), "threads": int(thu.value), "seconds": round(elu.value, 1)},
                "caveat": (f"this process may use {usable} of the host's {lim.get('os_cpu_count')} hardware threads (cgroup quota / affinity mask): the "
                           "figure is a baseline for THIS slice of the host, not for the whole host the >= 10x-at-8-GPUs target of "
                           "BASELINE.json is set against; per-core figure: per_core_T1"),
                "per_core_T1": round(t1, 3), "scaling_vs_T1": round(mt / t1, 1) if t1 > 0 else None,
                "host": lim}
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}


def reference_ratio(sample, preset, bcj, block_size):
    """Compressed size of the reference MT encoder on a few Blocks of the input (the ratio baseline)."""
    import _oracle as o
    nb = max(1, (len(sample) + block_size - 1) // block_size)
    enc = (o.ref_encode_mt_x86 if bcj else o.ref_encode_mt)(sample, preset, threads=min(nb, 16), block_size=block_size)
    return len(enc)


def host_to_host(host, preset, block_size, reps=3, bcj=False):
    """The SURVEY 8(d) end-to-end number: lzma_stream_encoder_mt + lzma_code(LZMA_FINISH) of libxz_amd.so on
    HOST buffers -- staging copy, H2D, device encode, D2H of the Stream all inside the timed region -- over the
    WHOLE input, and the WHOLE output back through the reference's own (multi-threaded) decoder, compared by sha256."""
    import ctypes as C
    import hashlib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_lzma_code import Mt, Stream
    L = xz_amd.lib()
    n = host.size
    out = np.empty(n // 2 + (n >> 3) + (1 << 20), dtype=np.uint8)
    best = None
    times = []

    class Filter(C.Structure):
        _fields_ = [("id", C.c_uint64), ("options", C.c_void_p)]
    class OptLzma(C.Structure):           # lzma_options_lzma (api/lzma/lzma12.h:216-525)
        _fields_ = [("dict_size", C.c_uint32), ("preset_dict", C.c_void_p), ("preset_dict_size", C.c_uint32),
                    ("lc", C.c_uint32), ("lp", C.c_uint32), ("pb", C.c_uint32), ("mode", C.c_int),
                    ("nice_len", C.c_uint32), ("mf", C.c_int), ("depth", C.c_uint32), ("pad", C.c_uint8 * 64)]
    fl = None
    if bcj:
        po = xz_amd.preset_options(preset)
        lz = OptLzma(dict_size=po.dict_size, lc=po.lc, lp=po.lp, pb=po.pb, mode=po.mode, nice_len=po.nice_len,
                     mf=po.mf, depth=po.depth)
        fl = (Filter * 3)(Filter(4, None), Filter(0x21, C.cast(C.pointer(lz), C.c_void_p)), Filter(2 ** 64 - 1, None))
    for _ in range(reps):
        s = Stream()
        m = Mt(threads=1, preset=preset, check=4, block_size=block_size)
        if fl is not None:
            m.filters = C.cast(fl, C.c_void_p)
        if L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) != 0:
            return None
        s.next_in = host.ctypes.data
        s.avail_in = n
        s.next_out = out.ctypes.data
        s.avail_out = out.size
        t0 = time.perf_counter()
        rc = L.lzma_code(C.byref(s), 3)
        while rc == 0:
            rc = L.lzma_code(C.byref(s), 3)
        dt = time.perf_counter() - t0
        total_out = s.total_out
        L.lzma_end(C.byref(s))
        if rc != 1:
            return {"value": None, "error": int(rc)}
        times.append(dt)
        if best is None or dt < best[0]:
            best = (dt, total_out)
    warm = sorted(times[1:]) if len(times) > 1 else sorted(times)          # (the first run allocates the context's work set)
    res = {"value": round(n / best[0] / 1e6, 2), "unit": "MB/s", "bytes": int(n), "ms": round(best[0] * 1e3, 1),
           "ratio": round(best[1] / n, 5),
           "mean_value": round(n / (sum(times) / len(times)) / 1e6, 2), "runs_ms": [round(t * 1e3, 1) for t in times],
           "median_warm_value": round(n / warm[len(warm) // 2] / 1e6, 2),
           "what": "lzma_stream_encoder_mt + lzma_code(LZMA_FINISH) of libxz_amd.so, the WHOLE input and output in host RAM "
                   "(staging, H2D, device encode, D2H inside the timed region), best of %d" % reps}
    try:
        import _oracle as o
        if o.have_ref():
            L.xzamd_release_parked()
            dec = np.empty(n + 16, dtype=np.uint8)
            t0 = time.perf_counter()
            r, dn = o.ref_decode_mt(out[:best[1]], dec)
            td = time.perf_counter() - t0
            ok = r == 1 and dn == n and hashlib.sha256(dec[:n]).digest() == hashlib.sha256(host).digest()
            res["roundtrip_reference_decoder_whole_output"] = {
                "ok": bool(ok), "bytes": int(n), "seconds": round(td, 1),
                "what": "the whole Stream through liblzma 5.8.3's lzma_stream_decoder_mt (oracle/_ref), sha256 of the result == sha256 of the input"}
    except Exception as e:  # noqa: BLE001
        res["roundtrip_reference_decoder_whole_output"] = {"ok": False, "error": str(e)}
    return res


def extra_configs():
    """BASELINE.json's other configurations as far as one GPU runs them, each as a child `bench.py` (its own process, its own
    device context), summarised: C2 = configs[1] (preset 1, 1 GiB synthetic text, 64 Blocks of 16 MiB), C4_1gpu = one GPU's
    share of configs[3] (preset 6, 4 GiB of a tar stream of the box's source trees), C5_1gpu = one GPU's run of configs[4]
    (preset 9e + x86 BCJ, 8 GiB of the box's ELF objects, 192 MiB Blocks)."""
    import subprocess
    runs = {
        "C2": ["--preset", "1", "--size-mib", "1024", "--block-mib", "16", "--steps", "3", "--warmup", "1", "--ratio-blocks", "16"],
        "C4_1gpu": ["--preset", "6", "--corpus", "tar", "--size-mib", "4096", "--steps", "2", "--warmup", "1", "--ratio-blocks", "4",
                    "--ratio-async", "--no-host-to-host"],
        "C5_1gpu": ["--preset", "0x80000009", "--bcj", "--corpus", "elf", "--size-mib", "8192", "--steps", "1", "--warmup", "1",
                    "--ratio-blocks", "1", "--ratio-async", "--no-host-to-host"],
    }
    out = {}
    for name, extra in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extra-configs", "--no-cpu-baseline"] + extra
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                out[name] = {"value": None, "error": (p.stderr or "")[-400:], "rc": p.returncode}
                continue
            d = json.loads(line[-1])
            h2h = d.get("host_to_host") or {}
            out[name] = {
                "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "workload": d["config"]["workload"], "data": d["data"], "corpus_sha256": d["config"].get("corpus_sha256"),
                "ratio": d["ratio"].get("ours"), "size_vs_reference_pct": d["ratio"].get("size_vs_reference_pct"),
                "ratio_sample_mib": d["ratio"].get("sample_mib"),
                "roundtrip_reference_decoder": d.get("roundtrip_reference_decoder"),
                "host_to_host": h2h.get("value"),
                "host_to_host_roundtrip_whole_output": (h2h.get("roundtrip_reference_decoder_whole_output") or {}).get("ok"),
                "roofline": d.get("roofline"), "stage_ms_last_step": d.get("stage_ms_last_step"),
                "wall_s": round(time.perf_counter() - t0, 1),
                "cmd": "python bench.py " + " ".join(extra),
            }
        except Exception as e:  # noqa: BLE001
            out[name] = {"value": None, "error": str(e)}
    return out


def ref_size_child(size_mib, preset, block_size, threads):
    """Child of the default run (`--ref-size-child`): the reference MT encoder over the WHOLE headline input (the same
    seeded corpus) on `threads` host threads at low priority, while the parent's other child runs keep the GPU busy;
    prints {"ref_bytes", "seconds", "threads"}.  No GPU, no torch."""
    import ctypes as C
    import numpy as np
    import _oracle as o
    try:
        os.nice(19)
    except OSError:
        pass
    n = size_mib << 20
    host = xz_amd.corpus_text(n, seed=1000)
    cap = n // 2 + (64 << 20)
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_size_t(0)
    f = o.ref().ref_encode_mt
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    t0 = time.perf_counter()
    r = f(host.ctypes.data, n, preset, threads, block_size, 4, out.ctypes.data, cap, C.byref(got))
    print(json.dumps({"rc": int(r), "ref_bytes": int(got.value), "seconds": round(time.perf_counter() - t0, 1), "threads": threads,
                      "in_bytes": n}), flush=True)
    return 0 if r == 1 else 1


def relaunch_under_torchrun(args_list, n):
    """`python bench.py --gpus N` with N > 1: spawn the N ranks ourselves (one process per GPU, RCCL)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args_list
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-mib", type=int, default=4096, help="input MiB per GPU (weak) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default; BASELINE's metric: 4 GiB at 1/2/4/8 GPUs): --size-mib in total, whole Blocks dealt to "
                         "the ranks in order; weak: --size-mib per GPU (BASELINE config 4's shape).  With more than one GPU the "
                         "line also carries one untimed-warm pass of the other mode (`other_scaling`)")
    ap.add_argument("--no-ratio", action="store_true", help="skip the ratio sample and the round trips (profiling runs)")
    ap.add_argument("--preset", type=lambda v: int(v, 0), default=6, help="0-9, | 0x80000000 for -e")
    ap.add_argument("--span-kib", type=int, default=0, help="0 = library default")
    ap.add_argument("--block-mib", type=int, default=0, help="0 = lzma_mt_block_size of the preset (BASELINE configs[1]: 16)")
    ap.add_argument("--parser", choices=["default", "fast", "optimal"], default="default",
                    help="device parser override (default: what the preset maps to)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--bcj", action="store_true", help="chain {x86 BCJ, LZMA2} (SURVEY.md 8d config C5)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="the default 1-GPU run also runs BASELINE's other single-GPU configurations (C2: preset 1, 1 GiB text, "
                         "16 MiB Blocks; C5_1gpu: 9e + x86 BCJ, 8 GiB ELF) as child processes and embeds their results as `configs`; "
                         "this switch leaves them out")
    ap.add_argument("--ratio-async", action="store_true",
                    help="start the reference encoder of the ratio sample on the host cores BEFORE the warm-up and collect it after "
                         "the timed steps (the child runs of `configs`: a 192 MiB Block at 9e costs liblzma two minutes on one "
                         "core; the device-resident path needs no CPU meanwhile)")
    ap.add_argument("--ratio-blocks", type=int, default=0,
                    help="Blocks of the input the ratio is measured on against the reference encoder (0 = about 1 GiB for the "
                         "headline workload, 4 Blocks otherwise)")
    ap.add_argument("--stream-sha", action="store_true", help="sha256 of the complete .xz Stream of the last step in the line")
    ap.add_argument("--ref-size-child", type=int, default=0, metavar="THREADS",
                    help="(internal) only run the reference encoder over the whole text corpus of --size-mib on THREADS host threads and print its size")
    ap.add_argument("--corpus", choices=["text", "elf", "tar"], default="text",
                    help="text: seeded synthetic enwik-style text (the headline workload); elf: the x86-64 shared "
                         "objects present on the box, concatenated and cycled (config C5's input); tar: ustar stream "
                         "of the box's source trees, cycled with a per-cycle perturbation (config C4's input)")
    args = ap.parse_args()

    if args.ref_size_child:
        o_ = xz_amd.preset_options(args.preset)
        sys.exit(ref_size_child(args.size_mib, args.preset, (args.block_mib << 20) if args.block_mib else xz_amd.mt_block_size(o_),
                                args.ref_size_child))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_under_torchrun(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # XZAMD_BENCH_OVERSUBSCRIBE=1: more ranks than GPUs (a 1-GPU box exercising the N-rank code path): the ranks share the
    # devices round robin and the gather runs over gloo on host tensors (RCCL refuses two ranks on one device)
    oversub = os.environ.get("XZAMD_BENCH_OVERSUBSCRIBE") == "1" and local_rank >= torch.cuda.device_count()
    oversub = oversub or (os.environ.get("XZAMD_BENCH_OVERSUBSCRIBE") == "1" and world > torch.cuda.device_count())
    dev_index = local_rank % torch.cuda.device_count() if oversub else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversub:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus

    opts = xz_amd.preset_options(args.preset)
    if args.parser != "default":
        opts.gpu_parser = 1 if args.parser == "optimal" else 0
        if not opts.gpu_parser:
            opts.gpu_sa_window = 0
    block_size = (args.block_mib << 20) if args.block_mib else xz_amd.mt_block_size(opts)
    total = args.size_mib << 20
    if args.scaling == "strong":
        nblocks = (total + block_size - 1) // block_size
        lo, hi = parallel.shard_blocks(nblocks, rank, world)
        n = min(total, hi * block_size) - lo * block_size if hi > lo else 0
    else:
        n = total
    # span size: derived by the library from the batch geometry and the GPU's wave slots (full rounds of
    # wavefronts per launch, between half the default and the default span) unless given
    opts.span_size = (args.span_kib << 10) if args.span_kib else xz_amd.SPAN_AUTO

    if args.bcj:
        opts.bcj = xz_amd.BCJ_X86
    # Strong scaling: every rank cuts ITS Blocks out of the ONE corpus the single-GPU run encodes (same generator, same seed),
    # so the N-rank Stream is byte for byte the 1-rank Stream (--stream-sha; tests/test_gpu_multirank.py compares them).  Weak
    # scaling: a corpus of its own per rank.
    first = lo * block_size if args.scaling == "strong" and world > 1 else 0
    seed = 1000 if args.scaling == "strong" else 1000 + rank
    if args.corpus == "elf":
        host = corpus_elf(max(first + n, 1), 0 if args.scaling == "strong" else rank)
    elif args.corpus == "tar":
        host = xz_amd.corpus_tar(max(first + n, 1), seed=seed)
    else:
        host = xz_amd.corpus_text(max(first + n, 1), seed=seed)
    host = host[first:first + n]
    import hashlib
    corpus_sha = hashlib.sha256(memoryview(host)).hexdigest() if rank == 0 and args.corpus != "text" else None   # text: a function of the seed
    data = torch.from_numpy(host).to(dev)
    enc = xz_amd.Encoder(dev_index)
    out_buf = torch.empty(xz_amd.lib().xzamd_stream_buffer_bound(n, block_size) + 64, dtype=torch.uint8, device=dev)

    def ratio_sample_bytes():
        # the headline workload: >= 1 GiB of it (43 Blocks of 24 MiB; the reference needs ~40 s of the box's 16 CPUs for
        # that); other workloads: 4 Blocks unless asked
        headline = args.preset == 6 and args.corpus == "text" and not args.bcj and not args.block_mib
        rb = args.ratio_blocks if args.ratio_blocks else (43 if headline else 4)
        return min(n, rb * block_size)

    ref_future = None
    if args.ratio_async and world == 1 and n and not args.no_ratio:
        import concurrent.futures as cf
        import _oracle as o_
        if o_.have_ref():
            ref_future = cf.ThreadPoolExecutor(1).submit(reference_ratio, host[:ratio_sample_bytes()], args.preset, args.bcj, block_size)

    def step():
        if world == 1:
            out, binfo = enc.encode(data, opts=opts, block_size=block_size, out=out_buf)
            return out, binfo
        out, binfo = enc.encode(data, opts=opts, block_size=block_size, out=out_buf, blocks_only=True)
        stream = parallel.gather_stream(out.cpu() if oversub else out, binfo, check=xz_amd.CHECK_CRC64)
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
        # the dominant kernel alone: the parse pieces (two-phase) or the single-phase span kernel (finder, span plan, seed
        # pieces and range coder are timed separately)
        # (two-phase: the FULL parse -- iteration 2 of k_parse_pieces -- without the partial iteration and the carried walk)
        enc_ms += st.ms_parse - st.ms_iter1 if two_phase(opts) else st.ms_encode - st.ms_find - st.ms_plan
        launches += st.encode_launches
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if oversub else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    st = enc.stats()
    other = None
    if world > 1:
        # one pass of the OTHER scaling mode (report both: the metric's strong form and config 4's weak form)
        o_mode = "weak" if args.scaling == "strong" else "strong"
        if o_mode == "strong":
            nb2 = (total + block_size - 1) // block_size
            lo2, hi2 = parallel.shard_blocks(nb2, rank, world)
            n2 = min(total, hi2 * block_size) - lo2 * block_size if hi2 > lo2 else 0
        else:
            n2 = total
        del data, out_buf
        torch.cuda.empty_cache()
        host2 = (corpus_elf(max(n2, 1), rank) if args.corpus == "elf" else
                 xz_amd.corpus_tar(max(n2, 1), seed=2000 + rank) if args.corpus == "tar" else xz_amd.corpus_text(max(n2, 1), seed=2000 + rank))[:n2]
        data = torch.from_numpy(host2).to(dev)
        out_buf = torch.empty(xz_amd.lib().xzamd_stream_buffer_bound(n2, block_size) + 64, dtype=torch.uint8, device=dev)
        step()                                   # warm
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e2 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cpu" if oversub else dev)
        dist.all_reduce(e2, op=dist.ReduceOp.MAX)
        jb2 = total if o_mode == "strong" else total * world
        other = {"scaling": o_mode, "value": round(jb2 / float(e2.item()) / 1e6, 2), "unit": "MB/s", "job_mib": jb2 >> 20,
                 "ms_per_step": round(float(e2.item()) * 1e3, 2), "steps": 1}
    job_bytes = total if args.scaling == "strong" else total * world
    total_in = job_bytes * args.steps
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": {"text": "synthetic", "elf": "x86-64 ELF shared objects of the box, concatenated/cycled",
                     "tar": "ustar stream of the box's source trees (" + xz_amd.TAR_ROOTS + "), cycled with a per-cycle perturbation"}[args.corpus],
            "config": {
                "workload": f"preset -{args.preset & 31}{'e' if args.preset >> 31 else ''} options (dict {opts.dict_size >> 20} MiB, {block_size >> 20} MiB Blocks, "
                            f"CRC64{', x86 BCJ + LZMA2' if args.bcj else ''}), {args.size_mib} MiB "
                            f"{ {'text': 'synthetic enwik-style text', 'elf': 'ELF shared objects', 'tar': 'tar stream of source trees'}[args.corpus]} "
                            f"{'per GPU' if args.scaling == 'weak' else 'in total, whole Blocks dealt to the ranks in order'}; `value` = input resident in HBM, "
                            f"output = complete .xz Stream in HBM; the same job through lzma_code with host buffers (SURVEY 8d end-to-end) is `host_to_host`",
                "world_size": world,
                "backend": (dist.get_backend() if world > 1 else None),
                "corpus_sha256": corpus_sha,       # of rank 0's input: ties "elf" / "tar" numbers to the image they were made on
                "device_match_finder": ((f"suffix-neighbourhood finder ({opts.gpu_sa_depth or 32}-byte-prefix suffix order, {opts.gpu_sa_window} slots per side + hash2/hash4 heads + equal 8/16 bytes)"
                                         if opts.gpu_sa_window else f"HC{opts.gpu_mf & 15} depth {opts.gpu_depth} (sort-built chains)")
                                        + f", nice {opts.gpu_nice_len}"),
                "device_parser": ("windowed optimal parser (384-node DP, exact prices, compound edges) over per-position match lists" if opts.gpu_parser
                                  else "lzma_lzma_optimum_fast semantics (greedy + 1-byte lazy)"),
                "span_bytes": int(st.span_size) if st.span_size else f"cost-balanced (work target {int(st.span_cost_used)} per span, >= 64 KiB)",
                "parallelism": (f"{world} x (two-phase: one wavefront per parse piece, {int(st.spans)} pieces on rank 0 incl. one 64 KiB seed piece per Block; "
                                f"one wavefront per encode span, {int(st.enc_spans)} encode spans walked in parallel with ONE continuous coder model per Block)" if two_phase(opts)
                                else f"{world} x (one wavefront per span, {int(st.spans)} spans on rank 0)"),
            },
            "ratio": {"ours": round(local_out_bytes / max(n, 1), 5)},
            **({"other_scaling": other} if other else {}),
            "roofline": {
                "bound": "hbm",
                "kernel": span_kernel_name(opts),
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": pmc_traffic(opts, args.corpus, args.preset, args.bcj)[0],
                "traffic_source": pmc_traffic(opts, args.corpus, args.preset, args.bcj)[1],
                "algorithmic_bytes_per_launch": int(alg_bytes / max(launches, 1)),
                "avg_launch_ms": round(enc_ms / max(launches, 1), 3),
                "launches": launches,
            },
            "stage_ms_last_step": {"chains": round(st.ms_chains, 2), "find": round(st.ms_find, 2),
                                   "span_plan": round(st.ms_plan, 2),
                                   **({"seed_pieces_under_the_finder": round(st.ms_seed, 2), "parse_pieces": round(st.ms_parse, 2),
                                       "of_which_partial_iteration_and_snapshots": round(st.ms_iter1, 2),
                                       "range_coder_second_stream": round(st.ms_code, 2)} if two_phase(opts)
                                      else {"span_encode": round(st.ms_encode - st.ms_find - st.ms_plan, 2)}),
                                   "crc": round(st.ms_crc, 2), "layout_and_assemble": round(st.ms_assemble, 2),
                                   "total": round(st.ms_total, 2)},
        }
        if args.stream_sha:
            res["stream_sha256"] = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()
            res["stream_bytes"] = int(out.numel())
        if world == 1 and n:
            import _oracle as o
            # ratio vs the reference on the same Blocks + bit-exact round trip through the REAL reference decoder
            try:
                if args.no_ratio:
                    pass
                elif o.have_ref():
                    sample_n = ratio_sample_bytes()
                    ref_size = ref_future.result() if ref_future is not None else reference_ratio(host[:sample_n], args.preset, args.bcj, block_size)
                    # (the span plan of a Block depends on the Block and the options only: the sample's Blocks are coded
                    # exactly as in the timed run)
                    s_out, _ = enc.encode(data[:sample_n], opts=opts, block_size=block_size)
                    res["ratio"]["sample_mib"] = sample_n >> 20
                    res["ratio"]["ours_on_sample"] = round(s_out.numel() / sample_n, 5)
                    res["ratio"][f"reference_xz_{args.preset & 31}{'e' if args.preset >> 31 else ''}_on_sample"] = round(ref_size / sample_n, 5)
                    res["ratio"]["size_vs_reference_pct"] = round(100.0 * (s_out.numel() / ref_size - 1), 2)
                    vn = min(sample_n, 64 << 20)
                    v_out, _ = enc.encode(data[:vn], opts=opts, block_size=block_size)
                    rr, dec = o.ref_decode(v_out.cpu().numpy().tobytes(), vn + 16)
                    res["roundtrip_reference_decoder"] = bool(rr == 1 and dec == host[:vn].tobytes())
                # the WHOLE job back through the device decoder (Block checks verified, bytes compared with the input)
                if not args.bcj and not args.no_ratio:
                    full, _ = enc.encode(data, opts=opts, block_size=block_size, out=out_buf)
                    torch.cuda.synchronize()
                    td = time.perf_counter()
                    dec_t, dnb = enc.decode(full, n, expected=data)
                    torch.cuda.synchronize()
                    td = time.perf_counter() - td
                    res["roundtrip_device_decoder"] = {"ok": bool(dec_t.numel() == n), "blocks": int(dnb),
                                                       "MB/s": round(n / td / 1e6, 1),
                                                       "what": "span-parallel verification decode of the whole Stream on the GPU + compare"}
                    del dec_t
            except Exception as e:  # noqa: BLE001
                res["roundtrip_reference_decoder"] = f"failed: {e}"
            if not args.no_host_to_host:
                try:
                    # the front end owns its own device context: give the device-resident one's work buffers back first
                    del data, out_buf, out
                    enc.close()
                    torch.cuda.empty_cache()
                    # (as many runs as timed steps, at most six: a warm context repeats to the millisecond)
                    res["host_to_host"] = host_to_host(host, args.preset, block_size, reps=max(3, min(args.steps, 6)), bcj=args.bcj)
                except Exception as e:  # noqa: BLE001
                    res["host_to_host"] = {"value": None, "error": str(e)}
            if not args.no_cpu_baseline:
                cb = cpu_baseline(host, args.preset, args.bcj, block_size)
                if cb is not None:
                    res["cpu_baseline"] = cb
            default_workload = (args.preset == 6 and args.corpus == "text" and not args.bcj and not args.block_mib
                                and args.size_mib == 4096 and args.parser == "default" and not args.span_kib)
            if default_workload and not args.no_extra_configs and not args.no_ratio:
                try:
                    del host
                    enc.close()
                    torch.cuda.empty_cache()
                    xz_amd.lib().xzamd_release_parked()
                except Exception:  # noqa: BLE001
                    pass
                # the ratio on the WHOLE job, not on a sample: the reference encodes all 4 GiB on the host cores (low priority,
                # three CPUs left to the children) underneath the child runs of `configs`, which keep the GPU busy meanwhile
                ref_child = None
                try:
                    import subprocess
                    if o.have_ref():
                        lim_ = cpu_limits()
                        ncpu = int(min(lim_.get("sched_getaffinity") or 16, lim_.get("cgroup_quota_cpus") or 1e9, os.cpu_count() or 16))
                        ref_child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ref-size-child", str(max(1, ncpu - 3)),
                                                      "--size-mib", str(args.size_mib), "--preset", str(args.preset)],
                                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                except Exception:  # noqa: BLE001
                    ref_child = None
                res["configs"] = extra_configs()
                if ref_child is not None:
                    try:
                        so, _ = ref_child.communicate(timeout=300)      # (it has had the minutes of the child runs already)
                        d_ = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
                        if d_.get("rc") == 1 and d_.get("in_bytes") == n:
                            res["ratio"]["whole_job"] = {
                                "ours_bytes": local_out_bytes, "reference_bytes": d_["ref_bytes"],
                                "ours": round(local_out_bytes / n, 5), "reference": round(d_["ref_bytes"] / n, 5),
                                "size_vs_reference_pct": round(100.0 * (local_out_bytes / d_["ref_bytes"] - 1), 2),
                                "what": (f"the complete .xz Stream of the timed job against liblzma 5.8.3 lzma_stream_encoder_mt (preset "
                                         f"{args.preset & 31}, same Block size) over the same {n >> 20} MiB, {d_['threads']} host threads at nice 19, "
                                         f"{d_['seconds']} s, run underneath the child runs of `configs`")}
                    except Exception as e:  # noqa: BLE001
                        res["ratio"]["whole_job"] = {"error": str(e)}
                        try:
                            ref_child.kill()
                        except Exception:  # noqa: BLE001
                            pass
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
