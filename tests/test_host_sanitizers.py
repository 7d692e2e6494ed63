"""Sanitizer builds of the plain-C host layer (ASan/UBSan builds, race detection).

xzamd_stream.c (lzma_code state machine, worker threads, ordered job queue) and xzamd_host.c (batch geometry, span
plan bookkeeping, layout, stored Blocks, framing) are compiled with -fsanitize=address,undefined and, separately,
-fsanitize=thread, on top of tests/host_stub/stub_xzk.c -- a CPU stand-in for the kernel layer whose "span kernel"
emits LZMA2 uncompressed chunks -- and driven through the liblzma entry points by tests/host_stub/driver.c.  The
Streams the host code frames must decode bit-exactly through the oracle decoder and the real reference decoder.
Test infrastructure only: nothing here is linked into libxz_amd.so."""
import os
import shutil
import subprocess

import pytest

import _oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "xz_amd", "csrc", f) for f in ("xzamd_stream.c", "xzamd_host.c", "corpus.c")] + \
      [os.path.join(ROOT, "tests", "host_stub", f) for f in ("stub_xzk.c", "driver.c")]


def _probe(flag, tmp):
    src = os.path.join(tmp, "p.c")
    open(src, "w").write("int main(void){return 0;}\n")
    exe = os.path.join(tmp, "p")
    r = subprocess.run(["gcc", flag, src, "-o", exe], capture_output=True)
    return r.returncode == 0 and subprocess.run([exe], capture_output=True).returncode == 0


@pytest.mark.parametrize("kind,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                                        ("tsan", ["-fsanitize=thread"])])
def test_host_layer_under_sanitizers(tmp_path, kind, flags):
    if not shutil.which("gcc") or not _probe(flags[0], str(tmp_path)):
        pytest.skip(f"{flags[0]} not usable here")
    exe = str(tmp_path / f"driver_{kind}")
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-pthread", *flags, *SRC, "-o", exe], check=True)
    outdir = tmp_path / "out"
    outdir.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([exe, str(outdir)], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "driver: ok" in p.stdout, (p.stdout[-2000:], p.stderr[-6000:])
    assert "runtime error" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr and "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[-6000:]
    expect_blocks = {"case1": 21, "case2": 4, "case3": 12, "case5": None, "case6": 0, "case7": 13, "case8": 21, "case9": 21,
                     "case10": 1, "case11": 2}
    for case, nb_want in expect_blocks.items():
        data = open(outdir / f"{case}.in", "rb").read()
        xz = open(outdir / f"{case}.xz", "rb").read()
        r, dec, nb = o.orc_xz_decode(xz, len(data) + 16)
        assert r == 0 and dec == data, (kind, case, r)
        if nb_want is not None:
            assert nb == nb_want, (kind, case, nb)
        if o.have_ref():
            rr, rdec = o.ref_decode(xz, len(data) + 16)
            assert rr == 1 and rdec == data, (kind, case)
