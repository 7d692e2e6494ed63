"""xz_amd -- MI355X-native LZMA2 Block encoder (Python plumbing over the C ABI).

The product is ``libxz_amd.so`` (HIP kernels for gfx950 + plain-C host layer,
see include/xz_amd.h).  This module only binds it with ctypes and uses torch
for device memory, streams and torch.distributed.  There is no CPU fallback:
if the library or a GPU is missing the calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XZ_AMD_LIB") or os.path.join(_HERE, "libxz_amd.so")   # env override: A/B builds

CHECK_NONE, CHECK_CRC32, CHECK_CRC64, CHECK_SHA256 = 0, 1, 4, 10
MF_HC3, MF_HC4, MF_BT4 = 0x03, 0x04, 0x14
PRESET_EXTREME = 0x80000000
SPAN_WHOLE_BLOCK = 0xFFFFFFFF
SPAN_DEFAULT = 0
SPAN_AUTO = 1
F_BLOCKS_ONLY = 1
BCJ_X86 = 4
BCJ_ARM64 = 0x0A
BCJ_RISCV = 0x0B
BCJ_POWERPC, BCJ_IA64, BCJ_ARM, BCJ_ARMTHUMB, BCJ_SPARC = 5, 6, 7, 8, 9


def filter_delta(dist):
    """xzamd_lzma_options.bcj value of a delta filter (dist 1..256) in front of LZMA2."""
    return 3 | ((dist - 1) << 8)



class LzmaOptions(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "dict_size", "lc", "lp", "pb", "mode", "nice_len", "mf", "depth",
        "gpu_mf", "gpu_nice_len", "gpu_depth", "span_size", "gpu_sa_window", "gpu_parser", "bcj",
        "gpu_sa_depth", "span_cost", "span_bits", "enc_span_bits", "bcj2", "bcj3", "part_iters")]


class Stats(C.Structure):
    _fields_ = [("in_bytes", C.c_uint64), ("out_bytes", C.c_uint64), ("blocks", C.c_uint64),
                ("spans", C.c_uint64), ("batches", C.c_uint64), ("blocks_stored", C.c_uint64),
                ("ms_chains", C.c_float), ("ms_encode", C.c_float), ("ms_crc", C.c_float),
                ("ms_assemble", C.c_float), ("ms_total", C.c_float), ("encode_launches", C.c_uint32),
                ("ms_find", C.c_float), ("span_size", C.c_uint32), ("ms_find_overlapped", C.c_float),
                ("span_cost_used", C.c_uint32), ("ms_plan", C.c_float), ("wave_slots", C.c_uint32),
                ("enc_spans", C.c_uint64), ("ms_seed", C.c_float), ("ms_parse", C.c_float), ("ms_code", C.c_float), ("ms_iter1", C.c_float)]


class BlockInfo(C.Structure):
    _fields_ = [("unpadded_size", C.c_uint64), ("uncompressed_size", C.c_uint64),
                ("out_offset", C.c_uint64), ("total_size", C.c_uint64)]


_lib = None


def lib():
    """Load libxz_amd.so (built in-tree by ``make -C xz_amd/csrc`` / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C xz_amd/csrc` "
                "(there is no CPU fallback for the encoder)")
        try:
            # torch brings its own copy of the HIP runtime; when libxz_amd.so (linked against /opt/rocm) is
            # loaded first, a later `import torch` puts a second runtime into the process and device
            # initialisation fails.  Loading torch first makes the linker reuse its runtime for ours.
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        l.xzamd_lzma_preset.argtypes = [C.POINTER(LzmaOptions), C.c_uint32]
        l.xzamd_mt_block_size.restype = C.c_uint64
        l.xzamd_mt_block_size.argtypes = [C.POINTER(LzmaOptions)]
        l.xzamd_block_buffer_bound.restype = C.c_uint64
        l.xzamd_block_buffer_bound.argtypes = [C.c_uint64]
        l.xzamd_stream_buffer_bound.restype = C.c_uint64
        l.xzamd_stream_buffer_bound.argtypes = [C.c_uint64, C.c_uint64]
        l.xzamd_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        l.xzamd_ctx_destroy.argtypes = [C.c_void_p]
        l.xzamd_ctx_set_batch_bytes.argtypes = [C.c_void_p, C.c_uint64]
        l.xzamd_last_error.restype = C.c_char_p
        l.xzamd_last_error.argtypes = [C.c_void_p]
        l.xzamd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        l.xzamd_stream_encode_device.argtypes = [
            C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(LzmaOptions), C.c_int,
            C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(BlockInfo),
            C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
        l.xzamd_frame_header.restype = C.c_uint64
        l.xzamd_frame_header.argtypes = [C.c_void_p, C.c_int]
        l.xzamd_frame_index_footer.restype = C.c_uint64
        l.xzamd_frame_index_footer.argtypes = [C.c_void_p, C.c_uint64, C.c_int,
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64]
        l.xzamd_stream_decode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                                 C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_uint64), C.c_void_p]
        l.xzamd_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        l.xzamd_trace_enable.argtypes = [C.c_void_p, C.c_uint32]
        l.xzamd_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        l.xzamd_corpus_lorem.argtypes = [C.c_void_p, C.c_uint64]
        l.xzamd_corpus_text.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
        l.xzamd_corpus_tar.restype = C.c_uint64
        l.xzamd_corpus_tar.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        l.xzamd_version.restype = C.c_char_p
        _lib = l
    return _lib


def preset_options(preset, span_size=SPAN_DEFAULT):
    o = LzmaOptions()
    if lib().xzamd_lzma_preset(C.byref(o), preset) != 0:
        raise ValueError(f"invalid preset {preset:#x}")
    o.span_size = span_size
    return o


def mt_block_size(opts):
    return lib().xzamd_mt_block_size(C.byref(opts))


def corpus_lorem(n):
    import numpy as np
    a = np.empty(n, dtype=np.uint8)
    lib().xzamd_corpus_lorem(a.ctypes.data, n)
    return a


def corpus_text(n, seed=1, threads=0):
    import numpy as np
    if threads <= 0:
        threads = min(64, os.cpu_count() or 1)
    a = np.empty(n, dtype=np.uint8)
    lib().xzamd_corpus_text(a.ctypes.data, n, seed, threads)
    return a


TAR_ROOTS = "/opt/rocm/include:/usr/include:/usr/lib/python3:/usr/lib/python3.10"


def corpus_tar(n, roots=TAR_ROOTS, seed=1):
    """ustar stream of the box's source trees, cycled to n bytes."""
    import numpy as np
    a = np.empty(n, dtype=np.uint8)
    if lib().xzamd_corpus_tar(a.ctypes.data, n, roots.encode(), seed) == 0:
        raise RuntimeError(f"no readable files under {roots}")
    return a


class XzAmdError(RuntimeError):
    pass


class Encoder:
    """One GPU context. ``encode`` takes/returns CUDA uint8 tensors (torch owns the memory)."""

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise XzAmdError("no GPU visible: the xz_amd encoder has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._ctx = C.c_void_p()
        rc = lib().xzamd_ctx_create(C.byref(self._ctx), self.device.index)
        if rc != 0:
            raise XzAmdError(f"xzamd_ctx_create failed: {rc}")

    def close(self):
        if self._ctx:
            lib().xzamd_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_batch_bytes(self, n):
        if lib().xzamd_ctx_set_batch_bytes(self._ctx, n) != 0:
            raise ValueError("batch bytes out of range")

    def stats(self):
        s = Stats()
        lib().xzamd_get_stats(self._ctx, C.byref(s))
        return s

    def encode(self, data, opts=None, preset=6, block_size=0, check=CHECK_CRC64,
               span_size=None, blocks_only=False, out=None):
        """Encode a CUDA uint8 tensor into an .xz Stream (CUDA uint8 tensor).

        Returns (out_tensor_view, block_infos). Runs on torch's current stream.
        """
        import torch
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        if opts is None:
            opts = preset_options(preset)
        if span_size is not None:
            opts.span_size = span_size
        n = data.numel()
        bs = block_size if block_size else mt_block_size(opts)
        cap = lib().xzamd_stream_buffer_bound(n, bs)
        if out is None or out.numel() < cap:
            out = torch.empty(cap, dtype=torch.uint8, device=data.device)
        nblocks = (n + bs - 1) // bs
        binfo = (BlockInfo * max(nblocks, 1))()
        out_size = C.c_uint64(0)
        nb = C.c_uint64(0)
        cur = torch.cuda.current_stream(data.device)
        stream = cur.cuda_stream
        if stream == 0:
            # The C ABI reads a NULL stream as "the context's own (non-blocking) stream", which is not ordered
            # after work queued on torch's legacy default stream: make the input final first.  The library
            # synchronises its stream before it returns, so the output is ordered for the caller either way.
            cur.synchronize()
        rc = lib().xzamd_stream_encode_device(
            self._ctx, C.c_void_p(data.data_ptr()), n, bs, C.byref(opts), check,
            F_BLOCKS_ONLY if blocks_only else 0, C.c_void_p(out.data_ptr()), out.numel(),
            C.byref(out_size), binfo, max(nblocks, 1), C.byref(nb), C.c_void_p(stream))
        if rc != 0:
            raise XzAmdError(f"xzamd_stream_encode_device failed ({rc}): "
                             f"{lib().xzamd_last_error(self._ctx).decode()}")
        return out[: out_size.value], list(binfo)[: nb.value]

    def decode(self, xz, out_cap, expected=None):
        """Decode an .xz Stream held in a CUDA uint8 tensor on the device.  With `expected` (CUDA uint8 tensor of
        the original data) it is a span-parallel verification decode.  Returns (decoded tensor view, nblocks)."""
        import torch
        assert xz.is_cuda and xz.dtype == torch.uint8 and xz.is_contiguous()
        if expected is not None:
            assert (expected.is_cuda and expected.dtype == torch.uint8 and expected.is_contiguous()
                    and expected.device == xz.device), "expected: contiguous CUDA uint8 tensor on the Stream's device"
        out = torch.empty(max(out_cap, 1), dtype=torch.uint8, device=xz.device)
        osz, mm, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        torch.cuda.current_stream(xz.device).synchronize()
        rc = lib().xzamd_stream_decode_device(
            self._ctx, C.c_void_p(xz.data_ptr()), xz.numel(), C.c_void_p(out.data_ptr()), out_cap, C.byref(osz),
            C.c_void_p(expected.data_ptr()) if expected is not None else None,
            expected.numel() if expected is not None else 0, C.byref(mm), C.byref(nb), None)
        if rc != 0:
            raise XzAmdError(f"xzamd_stream_decode_device failed ({rc}): {lib().xzamd_last_error(self._ctx).decode()}"
                             + (f" [{mm.value} mismatching words]" if mm.value else ""))
        return out[: osz.value], nb.value

    # debug hooks used by the parity tests
    def debug_fetch(self, what, count, dtype="uint32"):
        """Copy a work buffer of the last batch to the host: 1 = suffix order, 2 = position -> slot,
        3 = match-list records (8 x u32 per position), 4 = their u16 lengths."""
        import numpy as np
        buf = np.zeros(count, dtype=dtype)
        rc = lib().xzamd_debug_fetch(self._ctx, what, C.c_void_p(buf.ctypes.data), C.c_uint64(buf.nbytes))
        if rc != 0:
            raise XzAmdError(f"debug_fetch({what}) failed: {rc}")
        return buf

    def trace_enable(self, cap):
        if lib().xzamd_trace_enable(self._ctx, cap) != 0:
            raise XzAmdError("trace_enable failed")

    def trace_read(self, cap):
        import numpy as np
        buf = np.zeros((cap, 4), dtype=np.uint32)
        cnt = C.c_uint32(0)
        if lib().xzamd_trace_read(self._ctx, buf.ctypes.data, cap, C.byref(cnt)) != 0:
            raise XzAmdError("trace_read failed")
        return buf[: min(cnt.value, cap)], cnt.value
