#!/usr/bin/env python3
"""PCIe-inclusive rate: the liblzma streaming API of libxz_amd.so on HOST buffers (lzma_stream_encoder_mt +
lzma_code with LZMA_FINISH), i.e. staging H2D, device encode, D2H of the Stream.  Not the bench metric
(bench.py times the device-resident path); reported in DESIGN.md.

usage: tools/bench_lzma_code.py [MiB=1024] [preset=6] [reps=2]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import xz_amd  # noqa: E402


class Stream(C.Structure):
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_size_t), ("total_in", C.c_uint64),
                ("next_out", C.c_void_p), ("avail_out", C.c_size_t), ("total_out", C.c_uint64),
                ("allocator", C.c_void_p), ("internal", C.c_void_p),
                ("rp1", C.c_void_p), ("rp2", C.c_void_p), ("rp3", C.c_void_p), ("rp4", C.c_void_p),
                ("seek_pos", C.c_uint64), ("ri2", C.c_uint64), ("ri3", C.c_size_t), ("ri4", C.c_size_t),
                ("re1", C.c_int), ("re2", C.c_int)]


class Mt(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("threads", C.c_uint32), ("block_size", C.c_uint64),
                ("timeout", C.c_uint32), ("preset", C.c_uint32), ("filters", C.c_void_p),
                ("check", C.c_int), ("re1", C.c_int), ("re2", C.c_int), ("re3", C.c_int),
                ("ri1", C.c_uint32), ("ri2", C.c_uint32), ("ri3", C.c_uint32), ("ri4", C.c_uint32),
                ("memlimit_threading", C.c_uint64), ("memlimit_stop", C.c_uint64),
                ("ri7", C.c_uint64), ("ri8", C.c_uint64),
                ("rp1", C.c_void_p), ("rp2", C.c_void_p), ("rp3", C.c_void_p), ("rp4", C.c_void_p)]


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    preset = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    L = xz_amd.lib()
    n = mib << 20
    data = xz_amd.corpus_text(n, seed=1000)
    out = np.empty(n // 2 + (1 << 20), dtype=np.uint8)
    for r in range(reps):
        s = Stream()
        m = Mt(threads=1, preset=preset, check=4)
        assert L.lzma_stream_encoder_mt(C.byref(s), C.byref(m)) == 0
        s.next_in = data.ctypes.data
        s.avail_in = n
        s.next_out = out.ctypes.data
        s.avail_out = out.size
        t0 = time.perf_counter()
        rc = L.lzma_code(C.byref(s), 3)
        while rc == 0:
            rc = L.lzma_code(C.byref(s), 3)
        dt = time.perf_counter() - t0
        assert rc == 1, rc
        print(f"lzma_code host->host preset {preset} {mib} MiB: {dt*1e3:.1f} ms = {n/dt/1e6:.1f} MB/s, "
              f"ratio {s.total_out/n:.4f}", flush=True)
        L.lzma_end(C.byref(s))


if __name__ == "__main__":
    main()
