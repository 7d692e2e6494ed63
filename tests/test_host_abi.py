"""CPU-side checks of the product library: it loads without a GPU, exports every symbol the public
headers declare, and its host-side helpers (presets, sizes, framing) agree with the reference."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import _oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b((?:xzamd|lzma)_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


@pytest.mark.parametrize("header", ["xz_amd.h", "xz_amd_lzma.h"])
def test_exports_every_declared_symbol(product_lib, header):
    names = declared_functions(header)
    assert names, header
    missing = [n for n in names if not hasattr(product_lib, n)]
    assert not missing, missing


def test_presets_match_reference(product_lib):
    import xz_amd
    for preset in list(range(10)) + [p | xz_amd.PRESET_EXTREME for p in range(10)]:
        opt = xz_amd.preset_options(preset)
        if o.have_ref():
            r = (C.c_uint32 * 8)()
            assert o.ref().ref_preset(preset, r) == 0
            assert [opt.dict_size, opt.lc, opt.lp, opt.pb, opt.mode, opt.nice_len, opt.mf, opt.depth] == list(r), hex(preset)
            assert xz_amd.mt_block_size(opt) == o.ref().ref_mt_block_size_preset(preset)
        assert opt.gpu_mf in (xz_amd.MF_HC3, xz_amd.MF_HC4) and 1 <= opt.gpu_depth <= 56
    with pytest.raises(ValueError):
        xz_amd.preset_options(10)


def test_sizes_and_framing(product_lib):
    from xz_amd import parallel
    for u in (0, 1, 65535, 65536, 65537, 1 << 20, 24 << 20):
        assert product_lib.xzamd_block_buffer_bound(u) == o.orc().orc_block_bound(u)
    hdr = parallel.frame_header(4).tobytes()
    want = np.zeros(12, dtype=np.uint8)
    o.orc().orc_stream_header(o._ptr(want), 4)
    assert hdr == want.tobytes()
    # empty stream == header + empty index + footer
    empty = hdr + parallel.frame_index_footer([], [], 4).tobytes()
    r, dec, nb = o.orc_xz_decode(empty, 16)
    assert r == 0 and dec == b"" and nb == 0
    if o.have_ref():
        assert empty == o.ref_encode_mt(b"", 6, threads=2)


def test_corpus_generators(product_lib):
    import hashlib
    import xz_amd
    a = xz_amd.corpus_lorem(229001).tobytes()
    assert a == o.corpus_lorem(229001)
    # sha256 of the reference's generated test file
    assert hashlib.sha256(a).hexdigest().startswith("0e2490f0")
    t1 = xz_amd.corpus_text(3 << 20, seed=5, threads=1)
    t4 = xz_amd.corpus_text(3 << 20, seed=5, threads=4)
    assert (t1 == t4).all() and not (t1 == xz_amd.corpus_text(3 << 20, seed=6, threads=4)).all()


def test_no_gpu_means_error_not_fallback(product_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    assert product_lib.xzamd_ctx_create(C.byref(ctx), 0) != 0
    import xz_amd
    with pytest.raises(xz_amd.XzAmdError):
        xz_amd.Encoder()


def test_preload_interposer_falls_back_without_gpu(tmp_path):
    """libxz_amd_preload.so with the stock xz binary on a machine WITHOUT a usable device context must
    fall through to the real liblzma (preload.c): same bytes as plain xz, decodes, verbose note printed.
    (With a GPU the MT encoder is routed to the device: tests/test_gpu_parity.py.)"""
    import shutil
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked interposer test")
    xz = shutil.which("xz")
    pre = os.path.join(ROOT, "xz_amd", "libxz_amd_preload.so")
    if not xz or not os.path.exists(pre):
        pytest.skip("xz binary or preload library missing")
    data = (b"The quick brown fox jumps over the lazy dog. " * 40000)[:1500000]
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    env = dict(os.environ, LD_PRELOAD=pre, XZ_AMD_VERBOSE="1")
    p = subprocess.run([xz, "-T2", "-1", "-c", str(src)], capture_output=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    assert b"using liblzma" in p.stderr or b"cannot open" in p.stderr
    plain = subprocess.run([xz, "-T2", "-1", "-c", str(src)], capture_output=True, timeout=300)
    assert p.stdout == plain.stdout
    d = subprocess.run([xz, "-dc"], input=p.stdout, capture_output=True, env=env, timeout=300)
    assert d.returncode == 0 and d.stdout == data


def test_preload_library_exports_the_interposed_symbols():
    pre = os.path.join(ROOT, "xz_amd", "libxz_amd_preload.so")
    if not os.path.exists(pre):
        pytest.skip("preload library not built")
    L = C.CDLL(pre)
    for sym in ("lzma_stream_encoder_mt", "lzma_code", "lzma_end", "lzma_get_progress"):
        assert hasattr(L, sym), sym


REF_API = "/root/reference/src/liblzma/api"


def _real_header_dir():
    """The REAL liblzma headers: the reference tree's (dev container) or the system's liblzma-dev."""
    if os.path.exists(os.path.join(REF_API, "lzma.h")):
        return REF_API
    if os.path.exists("/usr/include/lzma.h"):
        return "/usr/include"
    return None


def test_client_compiles_against_the_real_lzma_h_and_links(tmp_path, product_lib):
    """examples/compress_mt.c compiled against the real <lzma.h> (no -DUSE_XZ_AMD) and linked against
    libxz_amd.so only: every liblzma symbol it uses must come from the drop-in library."""
    inc = _real_header_dir()
    if inc is None:
        pytest.skip("no real lzma.h on this box")
    exe = str(tmp_path / "compress_mt_real")
    subprocess.run(["gcc", "-O2", "-I" + inc, os.path.join(ROOT, "examples", "compress_mt.c"), "-o", exe,
                    "-L" + os.path.join(ROOT, "xz_amd"), "-lxz_amd", "-Wl,-rpath," + os.path.join(ROOT, "xz_amd"),
                    "-Wl,--no-undefined"], check=True)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "lzma_stream_encoder_mt" in undefined and "lzma_code" in undefined


def test_struct_layouts_equal_the_real_headers(tmp_path):
    """sizeof / offsetof of every struct that crosses the boundary, compiled once against the real <lzma.h>
    and once against include/xz_amd_lzma.h: must be identical (catches drift of the ABI restatement)."""
    inc = _real_header_dir()
    if inc is None:
        pytest.skip("no real lzma.h on this box")
    src = tmp_path / "probe.c"
    src.write_text(r'''
#ifdef USE_XZ_AMD
#include "xz_amd_lzma.h"
#else
#include <lzma.h>
#endif
#include <stddef.h>
#include <stdio.h>
#define O(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void)
{
    printf("lzma_stream %zu\nlzma_mt %zu\nlzma_options_lzma %zu\nlzma_filter %zu\nlzma_options_bcj %zu\nlzma_allocator %zu\n",
           sizeof(lzma_stream), sizeof(lzma_mt), sizeof(lzma_options_lzma), sizeof(lzma_filter),
           sizeof(lzma_options_bcj), sizeof(lzma_allocator));
    O(lzma_stream, next_in); O(lzma_stream, avail_in); O(lzma_stream, total_in); O(lzma_stream, next_out);
    O(lzma_stream, avail_out); O(lzma_stream, total_out); O(lzma_stream, allocator); O(lzma_stream, internal);
    O(lzma_stream, reserved_ptr1); O(lzma_stream, reserved_ptr4); O(lzma_stream, reserved_int2);
    O(lzma_stream, reserved_int3); O(lzma_stream, reserved_enum1); O(lzma_stream, reserved_enum2);
    O(lzma_mt, flags); O(lzma_mt, threads); O(lzma_mt, block_size); O(lzma_mt, timeout); O(lzma_mt, preset);
    O(lzma_mt, filters); O(lzma_mt, check);
    O(lzma_options_lzma, dict_size); O(lzma_options_lzma, preset_dict); O(lzma_options_lzma, preset_dict_size);
    O(lzma_options_lzma, lc); O(lzma_options_lzma, lp); O(lzma_options_lzma, pb); O(lzma_options_lzma, mode);
    O(lzma_options_lzma, nice_len); O(lzma_options_lzma, mf); O(lzma_options_lzma, depth);
    O(lzma_filter, id); O(lzma_filter, options); O(lzma_options_bcj, start_offset);
    printf("enums %d %d %d %d %d %d %d %d\n", (int)LZMA_STREAM_END, (int)LZMA_BUF_ERROR, (int)LZMA_PROG_ERROR,
           (int)LZMA_FULL_BARRIER, (int)LZMA_CHECK_CRC64, (int)LZMA_MODE_NORMAL, (int)LZMA_MF_BT4, (int)LZMA_FINISH);
    printf("ids %llx %llx\n", (unsigned long long)LZMA_FILTER_LZMA2, (unsigned long long)LZMA_FILTER_X86);
    return 0;
}
''')
    outs = []
    for flags in (["-I" + inc], ["-DUSE_XZ_AMD", "-I" + os.path.join(ROOT, "include")]):
        exe = str(tmp_path / ("probe" + str(len(outs))))
        subprocess.run(["gcc", "-O1", *flags, str(src), "-o", exe], check=True)
        outs.append(subprocess.run([exe], capture_output=True, text=True, check=True).stdout)
    assert outs[0] == outs[1], "\n".join(l for l in outs[0].splitlines() if l not in outs[1].splitlines())


def test_mt_block_size_of_filter_chains(product_lib):
    """lzma_mt_block_size (common/filter_encoder.c:270-292) for {LZMA2} and every {BCJ | delta, LZMA2} chain the
    device path takes: the same value as the real liblzma's."""
    import glob
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    reflib = C.CDLL(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "*.so"))[0])

    class Filter(C.Structure):
        _fields_ = [("id", C.c_uint64), ("options", C.c_void_p)]
    for lib in (product_lib, reflib):
        lib.lzma_mt_block_size.restype = C.c_uint64
        lib.lzma_mt_block_size.argtypes = [C.POINTER(Filter)]
    delta = (C.c_uint32 * 8)(0, 4)               # lzma_options_delta: type BYTE, dist 4
    for preset in (0, 3, 6, 9):
        lz = (C.c_uint8 * 128)()
        assert reflib.lzma_lzma_preset(lz, preset) == 0
        for fid in (None, 3, 4, 5, 6, 7, 8, 9, 0x0A, 0x0B):
            chain = (Filter * 3)()
            k = 0
            if fid is not None:
                chain[0].id, chain[0].options = fid, (C.cast(delta, C.c_void_p) if fid == 3 else None)
                k = 1
            chain[k].id, chain[k].options = 0x21, C.cast(lz, C.c_void_p)
            chain[k + 1].id, chain[k + 1].options = 0xFFFFFFFFFFFFFFFF, None
            want = reflib.lzma_mt_block_size(chain)
            assert want != 0 and product_lib.lzma_mt_block_size(chain) == want, (preset, fid)


def test_mt_block_size_error_cases_match_the_reference(product_lib):
    """filter_encoder.c:270-293 reports every failure as UINT64_MAX (clients test that value, src/xz/coder.c:482):
    NULL array, a filter id without an encoder, an invalid LZMA2 dictionary size, a chain in which no filter has a
    block size (BCJ / delta / LZMA1 only, or empty).  Same answers as the real liblzma."""
    import glob
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    reflib = C.CDLL(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "*.so"))[0])

    class Filter(C.Structure):
        _fields_ = [("id", C.c_uint64), ("options", C.c_void_p)]
    for lib in (product_lib, reflib):
        lib.lzma_mt_block_size.restype = C.c_uint64
        lib.lzma_mt_block_size.argtypes = [C.POINTER(Filter)]
    END = 0xFFFFFFFFFFFFFFFF
    lz = (C.c_uint8 * 128)()
    assert reflib.lzma_lzma_preset(lz, 6) == 0
    bad = (C.c_uint8 * 128)()
    assert reflib.lzma_lzma_preset(bad, 6) == 0
    C.cast(bad, C.POINTER(C.c_uint32))[0] = 100          # dict_size below LZMA_DICT_SIZE_MIN
    delta = (C.c_uint32 * 8)(0, 4)
    lzp, badp, dp = C.cast(lz, C.c_void_p), C.cast(bad, C.c_void_p), C.cast(delta, C.c_void_p)
    cases = {
        "empty": [(END, None)],
        "unknown id": [(0x77, None), (0x21, lzp), (END, None)],
        "unknown id behind LZMA2": [(0x21, lzp), (0x77, None), (END, None)],
        "bad dict": [(0x21, badp), (END, None)],
        "bcj only": [(4, None), (END, None)],
        "delta only": [(3, dp), (END, None)],
        "lzma1 only": [(0x4000000000000001, lzp), (END, None)],
        "five filters": [(4, None), (5, None), (3, dp), (7, None), (0x21, lzp), (END, None)],
    }
    for lib in (product_lib, reflib):
        assert lib.lzma_mt_block_size(None) == END
    for name, fl in cases.items():
        chain = (Filter * len(fl))()
        for i, (fid, opt) in enumerate(fl):
            chain[i].id, chain[i].options = fid, opt
        want = reflib.lzma_mt_block_size(chain)
        got = product_lib.lzma_mt_block_size(chain)
        assert got == want, (name, hex(got), hex(want))
        if name != "five filters":
            assert want == END, name
