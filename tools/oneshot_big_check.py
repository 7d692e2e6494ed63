#!/usr/bin/env python3
"""(GPU box) The one-shot API on an input far above the default Block size: lzma_easy_buffer_encode of N MiB of the bench text at a
preset -> how many Blocks the Stream holds (1 = the reference's layout; more = the fallback to the MT layout), time, size, and a round
trip through the reference's multi-threaded decoder.  usage: tools/oneshot_big_check.py [MiB=768] [preset=6]"""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xz_amd  # noqa: E402
import _oracle as o  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 768
preset = int(sys.argv[2], 0) if len(sys.argv) > 2 else 6
L = xz_amd.lib()
L.lzma_stream_buffer_bound.restype = C.c_size_t
L.lzma_stream_buffer_bound.argtypes = [C.c_size_t]
L.lzma_easy_buffer_encode.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t]
data = xz_amd.corpus_text(mib << 20, seed=1000)
out = C.create_string_buffer(L.lzma_stream_buffer_bound(data.size))
pos = C.c_size_t(0)
t0 = time.time()
r = L.lzma_easy_buffer_encode(preset, 4, None, data.ctypes.data, data.size, out, C.byref(pos), len(out))
dt = time.time() - t0
assert r == 0, r
xz = out.raw[:pos.value]
rr, dec, nb = o.orc_xz_decode(xz, data.size + 16)
ok = rr == 0 and hashlib.sha256(dec).digest() == hashlib.sha256(memoryview(data)).digest()
print(f"one-shot {mib} MiB preset {preset:#x}: {nb} Block(s), {pos.value} bytes (ratio {pos.value / data.size:.5f}), {dt:.2f} s incl. context creation, "
      f"decodes bit-exactly: {ok}")
sys.exit(0 if ok else 1)
