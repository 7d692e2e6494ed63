"""ctypes bindings for the test infrastructure under oracle/ (NOT product code).

* ``liboracle.so``      -- our plain-C restatements (container, LZMA2 decoder,
                            fast-mode encoder); always buildable with gcc.
* ``_ref/libref_shim.so`` -- flat wrappers around the REAL reference liblzma
                            5.8.3 compiled from /root/reference by oracle/Makefile
                            (prebuilt file travels to the GPU box).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def _ptr(a, typ=u8p):
    return a.ctypes.data_as(typ)


def build_oracle():
    """(Re)build liboracle.so; builds oracle/_ref too when the reference tree is here."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL)


def _load(path):
    if not os.path.exists(path):
        build_oracle()
    return C.CDLL(path)


class OrcParams(C.Structure):
    _fields_ = [("dict_size", C.c_uint32), ("lc", C.c_uint32), ("lp", C.c_uint32),
                ("pb", C.c_uint32), ("nice_len", C.c_uint32), ("mf", C.c_uint32),
                ("depth", C.c_uint32), ("span_size", C.c_uint32), ("sa_window", C.c_uint32),
                ("parser", C.c_uint32), ("sa_depth", C.c_uint32), ("span_cost", C.c_uint32),
                ("span_bits", C.c_uint32), ("enc_bits", C.c_uint32), ("part_iters", C.c_uint32)]


class OrcSymbol(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("back", C.c_uint32), ("len", C.c_uint32)]


class OrcTrace(C.Structure):
    _fields_ = [("sym", C.POINTER(OrcSymbol)), ("sym_cap", C.c_uint64),
                ("sym_count", C.c_uint64), ("chunks_lzma", C.c_uint64),
                ("chunks_uncompressed", C.c_uint64), ("state_resets", C.c_uint64),
                ("prop_resets", C.c_uint64), ("dict_resets", C.c_uint64)]


_orc = None
_ref = None


def orc():
    global _orc
    if _orc is None:
        lib = _load(os.path.join(ORACLE_DIR, "liboracle.so"))
        lib.orc_crc32.restype = C.c_uint32
        lib.orc_crc32.argtypes = [u8p, C.c_size_t, C.c_uint32]
        lib.orc_crc64.restype = C.c_uint64
        lib.orc_crc64.argtypes = [u8p, C.c_size_t, C.c_uint64]
        lib.orc_crc64_combine.restype = C.c_uint64
        lib.orc_crc64_combine.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        lib.orc_vli_encode.restype = C.c_uint32
        lib.orc_vli_encode.argtypes = [C.c_uint64, u8p]
        lib.orc_block_bound.restype = C.c_uint64
        lib.orc_block_bound.argtypes = [C.c_uint64]
        lib.orc_lzma2_dict_byte.restype = C.c_uint8
        lib.orc_lzma2_dict_byte.argtypes = [C.c_uint32]
        lib.orc_block_uncomp_encode.restype = C.c_uint64
        lib.orc_block_uncomp_encode.argtypes = [u8p, C.c_uint64, C.c_int, u8p, u64p]
        lib.orc_xz_frame.restype = C.c_uint64
        lib.orc_lzma2_decode.restype = C.c_int
        lib.orc_lzma2_decode.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, u64p,
                                         C.POINTER(OrcTrace)]
        lib.orc_xz_decode.restype = C.c_int
        lib.orc_xz_decode.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, u64p, u64p]
        lib.orc_lzma2_encode_block.restype = C.c_int
        lib.orc_lzma2_encode_block.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), u8p,
                                               C.c_uint64, u64p, C.POINTER(OrcTrace)]
        lib.orc_mf_dump.restype = C.c_int
        lib.orc_mf_dump.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), u32p, u32p,
                                    C.c_uint32, C.c_uint32, u32p, u32p, u32p]
        lib.orc_preset.restype = C.c_int
        lib.orc_preset.argtypes = [C.c_uint32, C.POINTER(OrcParams), u32p]
        _orc = lib
    return _orc


def have_ref():
    p = os.path.join(ORACLE_DIR, "_ref", "libref_shim.so")
    if not os.path.exists(p) and os.path.exists("/root/reference/src/liblzma/api/lzma.h"):
        build_oracle()
    return os.path.exists(p)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_shim.so"))
        lib.ref_encode_mt.restype = C.c_int
        lib.ref_encode_mt.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
                                      C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.ref_encode_mt_opts.restype = C.c_int
        lib.ref_encode_mt_opts.argtypes = [u8p, C.c_size_t] + [C.c_uint32] * 4 + [C.c_int, C.c_uint32, C.c_int, C.c_uint32,
                                           C.c_uint32, C.c_uint64, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.ref_raw_lzma2_encode.restype = C.c_int
        lib.ref_raw_lzma2_encode.argtypes = [u8p, C.c_size_t] + [C.c_uint32] * 4 + [C.c_int, C.c_uint32, C.c_int, C.c_uint32,
                                             u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.ref_decode.restype = C.c_int
        lib.ref_decode.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.ref_raw_lzma2_decode.restype = C.c_int
        lib.ref_raw_lzma2_decode.argtypes = [u8p, C.c_size_t, C.c_uint32, u8p, C.c_size_t,
                                             C.POINTER(C.c_size_t)]
        lib.ref_crc32.restype = C.c_uint32
        lib.ref_crc32.argtypes = [u8p, C.c_size_t, C.c_uint32]
        lib.ref_crc64.restype = C.c_uint64
        lib.ref_crc64.argtypes = [u8p, C.c_size_t, C.c_uint64]
        lib.ref_vli_encode.restype = C.c_int
        lib.ref_vli_encode.argtypes = [C.c_uint64, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.ref_block_buffer_bound.restype = C.c_uint64
        lib.ref_block_buffer_bound.argtypes = [C.c_uint64]
        lib.ref_mt_block_size_preset.restype = C.c_uint64
        lib.ref_mt_block_size_preset.argtypes = [C.c_uint32]
        lib.ref_preset.restype = C.c_int
        lib.ref_preset.argtypes = [C.c_uint32, u32p]
        lib.ref_block_uncomp_encode.restype = C.c_int
        lib.ref_block_uncomp_encode.argtypes = [u8p, C.c_size_t, C.c_int, u8p, C.c_size_t,
                                                C.POINTER(C.c_size_t), u64p]
        lib.ref_version.restype = C.c_char_p
        lib.ref_cputhreads.restype = C.c_uint32
        _ref = lib
    return _ref


# ---------------------------------------------------------------- helpers
def as_u8(data):
    if isinstance(data, (bytes, bytearray)):
        return np.frombuffer(bytes(data), dtype=np.uint8).copy()
    return np.ascontiguousarray(data, dtype=np.uint8)


def params_for_preset(preset, span_size=0):
    p = OrcParams()
    normal = C.c_uint32(0)
    assert orc().orc_preset(preset, C.byref(p), C.byref(normal)) == 0
    p.span_size = span_size
    return p, bool(normal.value)


def orc_encode_block(data, prm, want_trace=False):
    data = as_u8(data)
    cap = len(data) + len(data) // 8 + 4096
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    tr = None
    syms = None
    if want_trace:
        tr = OrcTrace()
        syms = (OrcSymbol * (len(data) + 16))()
        tr.sym = C.cast(syms, C.POINTER(OrcSymbol))
        tr.sym_cap = len(data) + 16
    r = orc().orc_lzma2_encode_block(_ptr(data), len(data), C.byref(prm), _ptr(out), cap,
                                     C.byref(n), C.byref(tr) if tr else None)
    assert r == 0, r
    res = out[: n.value].tobytes()
    if want_trace:
        s = np.frombuffer(syms, dtype=np.uint32).reshape(-1, 3)[: tr.sym_count].copy()
        return res, s, tr
    return res


def orc_decode_raw(payload, dict_size, out_cap, want_trace=False):
    payload = as_u8(payload)
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    n = C.c_uint64(0)
    tr = None
    syms = None
    if want_trace:
        tr = OrcTrace()
        syms = (OrcSymbol * (out_cap + 16))()
        tr.sym = C.cast(syms, C.POINTER(OrcSymbol))
        tr.sym_cap = out_cap + 16
    r = orc().orc_lzma2_decode(_ptr(payload), len(payload), dict_size, _ptr(out), out_cap,
                               C.byref(n), C.byref(tr) if tr else None)
    res = out[: n.value].tobytes()
    if want_trace:
        s = np.frombuffer(syms, dtype=np.uint32).reshape(-1, 3)[: min(tr.sym_count, tr.sym_cap)].copy()
        return r, res, s, tr
    return r, res


def orc_xz_decode(stream, out_cap):
    stream = as_u8(stream)
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    n = C.c_uint64(0)
    nb = C.c_uint64(0)
    r = orc().orc_xz_decode(_ptr(stream), len(stream), _ptr(out), out_cap, C.byref(n), C.byref(nb))
    return r, out[: n.value].tobytes(), nb.value


def ref_raw_encode(data, prm, mode=1):
    """Reference raw LZMA2 encoder with explicit options (mode 1 = fast, 2 = normal)."""
    data = as_u8(data)
    cap = len(data) + len(data) // 8 + 4096
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    mf = {3: 0x03, 4: 0x04}.get(prm.mf, prm.mf)
    r = ref().ref_raw_lzma2_encode(_ptr(data), len(data), prm.dict_size, prm.lc, prm.lp, prm.pb,
                                   mode, prm.nice_len, mf, prm.depth, _ptr(out), cap, C.byref(n))
    assert r == 1, r
    return out[: n.value].tobytes()


def ref_encode_mt(data, preset, threads=1, block_size=0, check=4):
    data = as_u8(data)
    cap = len(data) + len(data) // 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    r = ref().ref_encode_mt(_ptr(data), len(data), preset, threads, block_size, check,
                            _ptr(out), cap, C.byref(n))
    assert r == 1, r
    return out[: n.value].tobytes()


def ref_easy_buffer_encode(data, preset, check=4):
    """The reference's one-shot lzma_easy_buffer_encode: one Block whatever the size."""
    data = as_u8(data)
    cap = len(data) + len(data) // 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    f = ref().ref_easy_buffer_encode
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    r = f(_ptr(data), len(data), preset, check, _ptr(out), cap, C.byref(n))
    assert r == 0, r
    return out[: n.value].tobytes()


def ref_encode_mt_x86(data, preset, threads=1, block_size=0, check=4):
    """Reference MT encoder with the chain {x86 BCJ, LZMA2(preset)}."""
    data = as_u8(data)
    cap = len(data) + len(data) // 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    f = ref().ref_encode_mt_x86
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t,
                  C.POINTER(C.c_size_t)]
    r = f(_ptr(data), len(data), preset, threads, block_size, check, _ptr(out), cap, C.byref(n))
    assert r == 1, r
    return out[: n.value].tobytes()


def ref_encode_mt_chain(data, preset, filter_id, delta_dist=1, threads=1, block_size=0, check=4):
    """Reference MT encoder with the chain {filter_id, LZMA2(preset)} (x86 0x04, ARM64 0x0A, delta 0x03)."""
    data = as_u8(data)
    cap = len(data) + len(data) // 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    f = ref().ref_encode_mt_chain
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int,
                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    r = f(_ptr(data), len(data), preset, filter_id, delta_dist, threads, block_size, check, _ptr(out), cap, C.byref(n))
    assert r == 1, r
    return out[: n.value].tobytes()


def ref_encode_mt_chain_n(data, preset, chain, threads=1, block_size=0, check=4):
    """Reference MT encoder with up to three filters in front of LZMA2(preset): chain = [(filter_id, delta_dist), ...]."""
    data = as_u8(data)
    cap = len(data) + len(data) // 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    ids = (C.c_uint64 * 3)(*([c[0] for c in chain] + [0] * (3 - len(chain))))
    dists = (C.c_uint32 * 3)(*([c[1] for c in chain] + [0] * (3 - len(chain))))
    f = ref().ref_encode_mt_chain_n
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                  C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    r = f(_ptr(data), len(data), preset, len(chain), ids, dists, threads, block_size, check, _ptr(out), cap, C.byref(n))
    assert r == 1, r
    return out[: n.value].tobytes()


def ref_x86_filter(data):
    """What the reference's x86 BCJ encoder makes of one Block (fresh state, start offset 0)."""
    data = as_u8(data)
    out = np.empty(max(len(data), 1), dtype=np.uint8)
    f = ref().ref_x86_filter
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    r = f(_ptr(data), len(data), _ptr(out))
    assert r == 0, r
    return out[: len(data)].tobytes()


def orc_x86_encode(data):
    buf = np.array(as_u8(data), dtype=np.uint8, copy=True)
    f = orc().orc_x86_encode
    f.argtypes = [C.c_void_p, C.c_uint64]
    f.restype = None
    if len(buf):
        f(_ptr(buf), len(buf))
    return buf.tobytes()


def orc_sa_dump(data, depth=32):
    """(slot -> position, position -> slot) of one Block in the oracle's `depth`-byte-prefix suffix order."""
    data = as_u8(data)
    n = len(data)
    sa = np.zeros(max(n, 1), dtype=np.uint32)
    rk = np.zeros(max(n, 1), dtype=np.uint32)
    f = orc().orc_sa_dump
    f.restype = C.c_int
    f.argtypes = [u8p, C.c_uint32, C.c_uint32, u32p, u32p]
    assert f(_ptr(data), n, depth, _ptr(sa, u32p), _ptr(rk, u32p)) == 0
    return sa[:n], rk[:n]


def orc_list_dump(data, prm):
    """8 x u32 match-list record of every position of one Block (packed format), as the GPU stores it."""
    data = as_u8(data)
    n = len(data)
    w = np.zeros((max(n, 1), 8), dtype=np.uint32)
    f = orc().orc_list_dump
    f.restype = C.c_int
    f.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), u32p]
    assert f(_ptr(data), n, C.byref(prm), _ptr(w, u32p)) == 0
    return w[:n]


def corpus_x86(n, seed=1, density=24, runs=True):
    """Seeded x86-flavoured bytes: text-ish/zero/random background with CALL/JMP opcodes (E8/E9) whose
    rel32 has a 00/FF top byte, back-to-back opcodes, opcodes inside operands, and (runs=True) long
    stretches without any opcode-free 5-byte gap (worst case for chunk-parallel BCJ)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    a[rng.random(n) < 0.35] = 0
    k = 0
    while k + 8 < n:
        k += int(rng.integers(1, density))
        if k + 5 >= n:
            break
        a[k] = 0xE8 if rng.random() < 0.7 else 0xE9
        a[k + 4] = 0x00 if rng.random() < 0.5 else 0xFF
        if rng.random() < 0.15:
            a[k + 1 + int(rng.integers(0, 3))] = 0xE8
    if runs and n > 20000:
        s0 = n // 3
        a[s0:s0 + 9000:3] = 0xE8          # 9000 bytes without a synchronisation point
        a[s0 + 1:s0 + 9000:3] = 0xFF
    return a.tobytes()


def ref_raw_decode(payload, dict_size, out_cap):
    """Raw LZMA2 payload through the REAL reference decoder."""
    payload = as_u8(payload)
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    r = ref().ref_raw_lzma2_decode(_ptr(payload), len(payload), dict_size, _ptr(out), out_cap, C.byref(n))
    return r, out[: n.value].tobytes()


def ref_decode_mt(stream, out, threads=0):
    """Whole .xz Stream through the reference's MULTI-THREADED decoder into the numpy array `out`; returns
    (lzma_ret, bytes produced)."""
    stream = as_u8(stream) if not isinstance(stream, np.ndarray) else stream
    f = ref().ref_decode_mt
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t(0)
    r = f(stream.ctypes.data, stream.size, threads, out.ctypes.data, out.size, C.byref(n))
    return r, n.value


def ref_decode(stream, out_cap):
    stream = as_u8(stream)
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    n = C.c_size_t(0)
    r = ref().ref_decode(_ptr(stream), len(stream), _ptr(out), out_cap, C.byref(n))
    return r, out[: n.value].tobytes()


# ---------------------------------------------------------------- corpora
_LOREM = ("Lorem ipsum dolor sit amet, consectetur adipisicing elit, sed do eiusmod tempor "
          "incididunt ut labore et dolore magna aliqua. Ut enim ad minim veniam, quis nostrud "
          "exercitation ullamco laboris nisi ut aliquip ex ea commodo consequat. Duis aute irure "
          "dolor in reprehenderit in voluptate velit esse cillum dolore eu fugiat nulla pariatur. "
          "Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia deserunt "
          "mollit anim id est laborum.").split(" ")


def corpus_lorem(n):
    """tests/create_compress_files.c:110-152 (write_text) continued until n bytes."""
    assert len(_LOREM) == 69
    parts = []
    total = 0
    for w, word in enumerate(_LOREM):
        parts.append(word + " ")
        if w % 7 == 6:
            parts.append("\n")
    total = sum(len(p) for p in parts)
    x = 29
    while total < n:
        parts.append("\n\n")
        total += 2
        for w in range(69):
            x = (101771 * x + 71777) & 0xFFFFFFFF
            s = _LOREM[x % 69] + " "
            if w % 7 == 6:
                s += "\n"
            parts.append(s)
            total += len(s)
    return "".join(parts).encode("ascii")[:n]


def corpus_abc(n=49380):
    """tests/create_compress_files.c:82-88"""
    return (b"abc\n" * ((n + 3) // 4))[:n]


def corpus_random(n=493824):
    """tests/create_compress_files.c:93-105 (LCG, seed 5)"""
    k = (n + 3) // 4
    out = np.empty(k, dtype=np.uint32)
    x = 5
    for i in range(k):
        x = (101771 * x + 71777) & 0xFFFFFFFF
        out[i] = x
    return out.astype("<u4").tobytes()[:n]


def corpus_mixed(n, seed=1):
    """Seeded mix of text, repeats, binary-ish and random segments (edge-case stress)."""
    rng = np.random.default_rng(seed)
    lorem = corpus_lorem(min(n, 1 << 16))
    parts = []
    total = 0
    while total < n:
        kind = rng.integers(0, 5)
        ln = int(rng.integers(1, 5000))
        if kind == 0:
            off = int(rng.integers(0, max(1, len(lorem) - ln)))
            seg = lorem[off:off + ln]
        elif kind == 1:
            seg = bytes(rng.integers(0, 256, size=ln, dtype=np.uint8))
        elif kind == 2:
            seg = bytes([int(rng.integers(0, 256))]) * ln
        elif kind == 3:
            unit = bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8))
            seg = (unit * (ln // len(unit) + 1))[:ln]
        else:
            seg = bytes(rng.integers(0, 4, size=ln, dtype=np.uint8) + 97)
        parts.append(seg)
        total += len(seg)
    return b"".join(parts)[:n]


# ---------------------------------------------------------------- stream-level helpers
def orc_xz_stream(data, prm, block_size, check=4, payloads=None):
    """Whole .xz Stream in the reference MT layout from the oracle's per-Block payloads (computed here unless given)."""
    data = bytes(data)
    blocks = [data[i:i + block_size] for i in range(0, len(data), block_size)]
    if payloads is None:
        payloads = [orc_encode_block(b, prm) for b in blocks]
    nb = len(blocks)
    pay = [as_u8(p) for p in payloads]
    inp = [as_u8(b) for b in blocks]
    PP = (u8p * max(nb, 1))(*[_ptr(a) for a in pay])
    IP = (u8p * max(nb, 1))(*[_ptr(a) for a in inp])
    ps = (C.c_uint64 * max(nb, 1))(*[len(p) for p in payloads])
    isz = (C.c_uint64 * max(nb, 1))(*[len(b) for b in blocks])
    cap = len(data) + len(data) // 4 + 65536 + nb * 128
    out = np.empty(cap, dtype=np.uint8)
    n = orc().orc_xz_frame(PP, ps, IP, isz, C.c_uint64(nb), C.c_uint64(block_size),
                           C.c_uint32(prm.dict_size), C.c_int(check), _ptr(out), C.c_uint64(cap))
    return out[:n].tobytes()


def params_for_gpu_options(opts, span_size=None, span_cost_used=None):
    """Oracle parameters equal to what the device path runs for an xz_amd.LzmaOptions.  Cost-balanced spans
    (the default of the optimal-parser presets): span_cost_used = the target the device reports
    (Encoder.stats().span_cost_used; equals opts.span_cost unless the batch filled the GPU)."""
    p = OrcParams()
    p.dict_size = opts.dict_size
    p.lc, p.lp, p.pb = opts.lc, opts.lp, opts.pb
    p.nice_len = opts.gpu_nice_len
    p.mf = opts.gpu_mf & 0x0F
    p.depth = opts.gpu_depth
    p.sa_window = opts.gpu_sa_window
    p.parser = opts.gpu_parser
    p.sa_depth = opts.gpu_sa_depth
    sp = opts.span_size if span_size is None else span_size
    if sp in (0, 1) and opts.gpu_parser and opts.gpu_sa_window and opts.span_cost:
        p.span_size = 65536                         # XZAMD_SPAN_MIN_LEN
        p.span_cost = span_cost_used if span_cost_used else opts.span_cost
        p.span_bits = opts.span_bits
        p.enc_bits = opts.enc_span_bits                # != 0: two-phase (parse pieces + encode spans)
        p.part_iters = opts.part_iters
        return p
    if sp in (0, 1):
        sp = 131072 if opts.gpu_parser else 262144 if opts.dict_size >= (1 << 20) else 65536    # xzamd_host.c DEFAULT_SPAN_OPT / _FAST_BIG / DEFAULT_SPAN
    p.span_size = 0 if sp == 0xFFFFFFFF else sp
    return p


def orc_span_plan(data, prm):
    """(chunk work estimates, chunk bit estimates, span starts) of one Block under prm.span_cost."""
    data = as_u8(data)
    n = len(data)
    m = (n + 4095) // 4096
    cc = np.zeros(2 * max(m, 1), dtype=np.uint32)
    ss = np.zeros(n // 4096 + 2, dtype=np.uint32)
    f = orc().orc_span_plan
    f.restype = C.c_uint32
    f.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), u32p, u32p, C.c_uint32]
    ns = f(_ptr(data), n, C.byref(prm), _ptr(cc, u32p), _ptr(ss, u32p), len(ss))
    assert ns > 0
    return cc[:m], cc[m:2 * m], ss[:ns]


def orc_piece_plan(data, prm):
    """Two-phase plan of one Block: (piece starts, encode-span starts)."""
    data = as_u8(data)
    n = len(data)
    ss = np.zeros(n // 4096 + 2, dtype=np.uint32)
    es = np.zeros(n // 4096 + 2, dtype=np.uint32)
    ne = C.c_uint32(0)
    f = orc().orc_piece_plan
    f.restype = C.c_uint32
    f.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), u32p, u32p, C.c_uint32, u32p, C.c_uint32, u32p]
    ns = f(_ptr(data), n, C.byref(prm), None, _ptr(ss, u32p), len(ss), _ptr(es, u32p), len(es), C.byref(ne))
    assert ns > 0
    return ss[:ns], es[:ne.value]


def orc_parse_dump(data, prm):
    """Two-phase: the recorded parse of one Block (sym_len u16, sym_dist u32 per position, valid at symbol starts)."""
    data = as_u8(data)
    n = len(data)
    sl = np.zeros(n, dtype=np.uint16)
    sd = np.zeros(n, dtype=np.uint32)
    f = orc().orc_parse_dump
    f.restype = C.c_int
    f.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), C.POINTER(C.c_uint16), u32p]
    r = f(_ptr(data), n, C.byref(prm), _ptr(sl, C.POINTER(C.c_uint16)), _ptr(sd, u32p))
    assert r == 0, r
    return sl, sd


class OrcTwoPhaseDbg(C.Structure):
    _fields_ = [("snap_sr", u32p), ("snap_probs", C.POINTER(C.c_uint16)), ("price", C.POINTER(C.c_uint64)), ("carry", u32p)]


def orc_two_phase_debug(data, prm, npieces, nspans):
    """Two-phase, stage by stage (oracle.h: orc_two_phase_debug): per piece the state / rep distances (npieces x 5) and the 1846
    non-literal probabilities iteration 2 starts from, the parser's price of every piece; per encode span the carry decision."""
    data = as_u8(data)
    sr = np.zeros((npieces, 5), dtype=np.uint32)
    probs = np.zeros((npieces, 1846), dtype=np.uint16)
    price = np.zeros(npieces, dtype=np.uint64)
    carry = np.zeros(nspans, dtype=np.uint32)
    d = OrcTwoPhaseDbg(_ptr(sr, u32p), _ptr(probs, C.POINTER(C.c_uint16)), _ptr(price, C.POINTER(C.c_uint64)), _ptr(carry, u32p))
    f = orc().orc_two_phase_debug
    f.restype = C.c_int
    f.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), C.POINTER(OrcTwoPhaseDbg)]
    r = f(_ptr(data), len(data), C.byref(prm), C.byref(d))
    assert r == 0, r
    return sr, probs, price, carry


def orc_encode_block_syms(data, prm):
    """Two-phase: (raw LZMA2 payload, sym_len, sym_dist) of one Block in one pass of the oracle."""
    data = as_u8(data)
    n = len(data)
    cap = n + n // 8 + 4096
    out = np.empty(cap, dtype=np.uint8)
    sl = np.zeros(n, dtype=np.uint16)
    sd = np.zeros(n, dtype=np.uint32)
    sz = C.c_uint64(0)
    f = orc().orc_lzma2_encode_block_syms
    f.restype = C.c_int
    f.argtypes = [u8p, C.c_uint32, C.POINTER(OrcParams), u8p, C.c_uint64, u64p, C.POINTER(C.c_uint16), u32p]
    r = f(_ptr(data), n, C.byref(prm), _ptr(out), cap, C.byref(sz), _ptr(sl, C.POINTER(C.c_uint16)), _ptr(sd, u32p))
    assert r == 0, r
    return out[: sz.value].tobytes(), sl, sd


def first_diff(a, b):
    n = min(len(a), len(b))
    aa = np.frombuffer(a, dtype=np.uint8, count=n)
    bb = np.frombuffer(b, dtype=np.uint8, count=n)
    d = np.nonzero(aa != bb)[0]
    if len(d):
        return int(d[0])
    return -1 if len(a) == len(b) else n


def orc_set_log_cap(v):
    """Tests only: logged bits per encode span and probability of the carried model walk (0 = default); the device: XZAMD_TEST_LOG_CAP."""
    f = orc().orc_set_log_cap
    f.restype = None
    f.argtypes = [C.c_uint32]
    f(v)


def orc_set_tok_per_byte(v):
    """Tests only: token budget per input byte of the two-phase coder (0 = default); the device: XZAMD_TEST_TOK_PER_BYTE."""
    f = orc().orc_set_tok_per_byte
    f.restype = None
    f.argtypes = [C.c_uint32]
    f(v)
