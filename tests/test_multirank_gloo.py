"""N>1 path on CPU: world_size-2 gloo run of the Block sharding + gather + framing logic.
The encoded Blocks come from the oracle here (no GPU in this container); the framed result must
equal the single-process stream and decode back to the input."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 1 << 16


class _BI:
    def __init__(self, unp, unc):
        self.unpadded_size, self.uncompressed_size = unp, unc


def _blocks_for(data, lo, hi):
    """Framed Blocks lo..hi-1 of `data` (BS-byte Blocks) cut out of the oracle's whole stream."""
    prm, _ = o.params_for_preset(1)
    blob = b""
    infos = []
    for b in range(lo, hi):
        piece = data[b * BS:(b + 1) * BS]
        s = o.orc_xz_stream(piece, prm, BS)        # header(12) + one Block + index + footer
        payload = o.orc_encode_block(piece, prm)
        hs = s[12] * 4 + 4
        total = hs + ((len(payload) + 3) & ~3) + 8
        blob += s[12:12 + total]
        infos.append(_BI(hs + len(payload) + 8, len(piece)))
    return blob, infos


def _worker(rank, world, port, data, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xz_amd import parallel
    nblocks = (len(data) + BS - 1) // BS
    lo, hi = parallel.shard_blocks(nblocks, rank, world)
    blob, infos = _blocks_for(data, lo, hi)
    t = torch.frombuffer(bytearray(blob), dtype=torch.uint8) if blob else torch.empty(0, dtype=torch.uint8)
    out = parallel.gather_stream(t, infos, check=4)
    if rank == 0:
        q.put(out.numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nbytes", [5 * BS + 123, BS // 2])
def test_gather_stream_world2(nbytes):
    data = o.corpus_lorem(nbytes)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if nbytes < BS else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prm, _ = o.params_for_preset(1)
    want = o.orc_xz_stream(data, prm, BS)
    assert got == want
    r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    assert r == 0 and dec == data


def test_shard_blocks_partition():
    from xz_amd import parallel
    for n in (0, 1, 7, 8, 171, 1366):
        for w in (1, 2, 4, 8):
            ranges = [parallel.shard_blocks(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
