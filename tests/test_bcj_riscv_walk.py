"""RISC-V BCJ (simple/riscv.c:352-609): the restatement behind k_riscv_bcj and the synchronisation rule that lets
it cut the reference's serial walk into chunks, checked on the CPU against the real filter
(lzma_bcj_riscv_encode of oracle/_ref).  The device kernel is the same logic in HIP; its own parity test
(tests/test_gpu_parity.py::test_simple_bcj_chains_identical_to_reference[riscv]) compares whole .xz Streams.

The rule: what the reference finds at an examined position decides how far it jumps (2, 4, 6 or 8 bytes), and it
reads only bytes no earlier conversion has touched, so step(i) is a function of the input.  A position is skipped
only when an examined position 2, 4 or 6 bytes before it jumps over it; therefore a position with
step(i-2) <= 2, step(i-4) <= 4 and step(i-6) <= 6 is examined by EVERY walk, whatever happened before.  Each
chunk's owner starts at the first such position inside its chunk and stops at the first one behind its chunk."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import _oracle as o

M = 0xFFFFFFFF


def _rd32(b, i):
    return b[i] | b[i + 1] << 8 | b[i + 2] << 16 | b[i + 3] << 24


def _not_pair(auipc, inst2):          # rd of the AUIPC != rs1 of the second instruction, or its opcode does not end in 11
    return (((auipc << 8) ^ inst2) & 0xF8003) != 3


def _special(auipc):                  # the form the encoder itself produces: rd = x2, bits 13:12 = 11, "rs1" not x0 / x2
    return (auipc & 0x3FFF) == 0x3117 and ((auipc >> 27) & 0x1D) != 0


def _step(b, i, limit):
    if i > limit:
        return 2
    b0 = b[i]
    if b0 == 0xEF:
        return 2 if (b[i + 1] & 0x0D) else 4
    if (b0 & 0x7F) != 0x17:
        return 2
    inst = _rd32(b, i)
    if inst & 0xE80:
        return 6 if _not_pair(inst, _rd32(b, i + 4)) else 8
    return 8 if _special(inst) else 4


def _sync(b, i, limit):
    return ((i < 2 or _step(b, i - 2, limit) <= 2) and (i < 4 or _step(b, i - 4, limit) <= 4)
            and (i < 6 or _step(b, i - 6, limit) <= 6))


def _convert(b, out, pos):
    """Examine position pos: convert into `out` when the reference would, return the jump length."""
    b0 = b[pos]
    if b0 == 0xEF:
        b1 = b[pos + 1]
        if b1 & 0x0D:
            return 2
        b2, b3 = b[pos + 2], b[pos + 3]
        addr = (((b1 & 0xF0) << 8) | ((b2 & 0x0F) << 16) | ((b2 & 0x10) << 7) | ((b2 & 0xE0) >> 4)
                | ((b3 & 0x7F) << 4) | ((b3 & 0x80) << 13))
        addr = (addr + pos) & M
        out[pos + 1] = (b1 & 0x0F) | ((addr >> 13) & 0xF0)
        out[pos + 2] = (addr >> 9) & 0xFF
        out[pos + 3] = (addr >> 1) & 0xFF
        return 4
    if (b0 & 0x7F) != 0x17:
        return 2
    inst = _rd32(b, pos)
    if inst & 0xE80:
        inst2 = _rd32(b, pos + 4)
        if _not_pair(inst, inst2):
            return 6
        addr = ((inst & 0xFFFFF000) + (inst2 >> 20) - ((inst2 >> 19) & 0x1000) + pos) & M
        inst = (0x17 | (2 << 7) | (inst2 << 12)) & M
        out[pos:pos + 4] = list(inst.to_bytes(4, "little"))
        out[pos + 4:pos + 8] = list(addr.to_bytes(4, "big"))
        return 8
    if not _special(inst):
        return 4
    fake_addr = _rd32(b, pos + 4)
    fake_inst2 = ((inst >> 12) | (fake_addr << 20)) & M
    inst = (0x17 | ((inst >> 27) << 7) | (fake_addr & 0xFFFFF000)) & M
    out[pos:pos + 4] = list(inst.to_bytes(4, "little"))
    out[pos + 4:pos + 8] = list(fake_inst2.to_bytes(4, "little"))
    return 8


def serial_walk(data):
    b, out = list(data), list(data)
    if len(b) >= 8:
        limit, pos = len(b) - 8, 0
        while pos <= limit:
            pos += _convert(b, out, pos)
    return bytes(out)


def chunked_walk(data, chunk):
    """What k_riscv_bcj does, one loop iteration per chunk owner (owners are independent of each other)."""
    b, out = list(data), list(data)
    if len(b) < 8:
        return bytes(out)
    limit = len(b) - 8
    for k in range((len(b) + chunk - 1) // chunk):
        s = k * chunk
        if s > limit:
            continue
        e = s + chunk
        pos = 0
        if k:
            q = s
            while q <= limit and q < e and not _sync(b, q, limit):
                q += 2
            if not (q <= limit and q < e):
                continue                       # no synchronisation point here: the previous owner walks through
            pos = q
        while pos <= limit:
            if pos >= e and _sync(b, pos, limit):
                break                          # the next owner's start
            pos += _convert(b, out, pos)
    return bytes(out)


def riscv_like(n, seed):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 256, n + 16, dtype=np.uint8)
    for i in range(0, max(n - 8, 0), 2):
        r = rng.random()
        if r < 0.08:
            b[i] = 0xEF
            if rng.random() < 0.7:
                b[i + 1] &= 0xF2
        elif r < 0.2:
            inst = (int(rng.integers(0, 1 << 32)) & ~0x7F) | 0x17
            if rng.random() < 0.5:
                rd = int(rng.choice([1, 3, 5, 6, 10, 31]))
                inst = (inst & ~(0x1F << 7)) | (rd << 7)
                if rng.random() < 0.7:
                    i2 = (int(rng.integers(0, 1 << 32)) & ~(0x1F << 15)) | (rd << 15) | 3
                    b[i + 4:i + 8] = np.frombuffer(i2.to_bytes(4, "little"), dtype=np.uint8)
            else:
                inst = (inst & ~0x3FFF) | 0x3117 if rng.random() < 0.7 else (inst & ~(0x1F << 7)) | (int(rng.choice([0, 2])) << 7)
                if rng.random() < 0.3:
                    inst &= 0x07FFFFFF
            b[i:i + 4] = np.frombuffer((inst & M).to_bytes(4, "little"), dtype=np.uint8)
    return bytes(b[:n])


@pytest.fixture(scope="module")
def ref_filter():
    if not o.have_ref():
        pytest.skip("oracle/_ref not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(glob.glob(os.path.join(root, "oracle", "_ref", "*.so"))[0])
    lib.lzma_bcj_riscv_encode.restype = C.c_size_t
    lib.lzma_bcj_riscv_encode.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]

    def run(data):
        buf = C.create_string_buffer(bytes(data), max(len(data), 1))
        lib.lzma_bcj_riscv_encode(0, buf, len(data))
        return buf.raw[:len(data)]
    return run


@pytest.mark.parametrize("n", [0, 5, 8, 9, 17, 100, 1000, 4097, 12000])
def test_restatement_and_chunk_rule_equal_the_reference_filter(ref_filter, n):
    converted = 0
    for seed in range(3):
        data = riscv_like(n, 100 * n + seed)
        want = ref_filter(data)
        converted += sum(x != y for x, y in zip(data, want))
        assert serial_walk(data) == want, ("serial restatement", n, seed)
        for chunk in (16, 64, 2048):
            assert chunked_walk(data, chunk) == want, ("chunk owners", n, seed, chunk)
    if n >= 1000:
        assert converted > n // 20            # the generator does plant convertible instructions
