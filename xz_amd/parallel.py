"""Multi-GPU plumbing: Blocks are independent (doc/faq.txt:156-196 of the reference), so the input
is sharded by whole Blocks across ranks with NO data-path collective; the only exchange step is the
gather of the encoded Blocks (+ their 16-byte Index records) to the rank that frames the Stream --
torch.distributed send/recv, i.e. RCCL over xGMI for CUDA tensors (gloo in the CPU tests).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import lib


def shard_blocks(nblocks, rank, world):
    """Contiguous Block range [lo, hi) of `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(nblocks, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frame_header(check):
    buf = np.zeros(12, dtype=np.uint8)
    lib().xzamd_frame_header(buf.ctypes.data, check)
    return buf


def frame_index_footer(unpadded, uncompressed, check):
    n = len(unpadded)
    cap = 64 + 18 * n
    buf = np.zeros(cap, dtype=np.uint8)
    a = (C.c_uint64 * max(n, 1))(*[int(x) for x in unpadded])
    b = (C.c_uint64 * max(n, 1))(*[int(x) for x in uncompressed])
    w = lib().xzamd_frame_index_footer(buf.ctypes.data, cap, check, a, b, n)
    assert w > 0
    return buf[:w]


def gather_stream(blocks, binfo, check, dst=0, group=None):
    """blocks: 1-D uint8 tensor holding this rank's encoded Blocks back to back (any device);
    binfo: sequence with .unpadded_size/.uncompressed_size per Block.  Returns the complete .xz
    Stream (same device) on rank `dst`, None elsewhere."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    dev = blocks.device
    meta = torch.tensor([blocks.numel(), len(binfo)], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    sizes = [int(m[0].item()) for m in metas]
    counts = [int(m[1].item()) for m in metas]
    maxb = max(max(counts), 1)
    rec = torch.zeros((maxb, 2), dtype=torch.int64)
    for i, b in enumerate(binfo):
        rec[i, 0] = int(b.unpadded_size)
        rec[i, 1] = int(b.uncompressed_size)
    rec = rec.to(dev)
    recs = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(recs, rec, group=group)
    if rank != dst:
        if blocks.numel():
            dist.send(blocks, dst=dst, group=group)
        return None
    unp, unc = [], []
    for r in range(world):
        rr = recs[r].cpu()
        for i in range(counts[r]):
            unp.append(int(rr[i, 0]))
            unc.append(int(rr[i, 1]))
    tail = frame_index_footer(unp, unc, check)
    total = 12 + sum(sizes) + len(tail)
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    out[:12] = torch.from_numpy(frame_header(check)).to(dev)
    # all receives are posted up front (one per peer link of the xGMI mesh), then waited for together.  Each lands in a
    # staging tensor of its own -- allocator-aligned, whereas a peer's place in the Stream starts at an arbitrary multiple of
    # four bytes, which RCCL's send/recv kernels handle with narrow accesses at best -- and is copied into place on the device
    # (a device-to-device copy at HBM rate: nothing next to the link rate)
    off = 12
    pending = []
    for r in range(world):
        if sizes[r] == 0:
            continue
        if r == rank:
            out[off:off + sizes[r]] = blocks
        else:
            stage = torch.empty(sizes[r], dtype=torch.uint8, device=dev)
            pending.append((dist.irecv(stage, src=r, group=group), stage, off))
        off += sizes[r]
    out[off:] = torch.from_numpy(tail.copy()).to(dev)
    for req, stage, at in pending:
        req.wait()
        out[at:at + stage.numel()].copy_(stage)
    return out
