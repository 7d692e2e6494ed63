#!/usr/bin/env python3
"""Experiment: two encoder contexts on two streams, each taking half of the Blocks, driven by two host threads."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, xz_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 512
host = xz_amd.corpus_text(mib << 20, seed=1000)
t = torch.from_numpy(host).cuda()
opts = xz_amd.preset_options(6)
bs = xz_amd.mt_block_size(opts)
nblocks = (t.numel() + bs - 1) // bs
encs = [xz_amd.Encoder() for _ in range(lanes)]
streams = [torch.cuda.Stream() for _ in range(lanes)]
for e in encs:
    e.set_batch_bytes(batch_mib << 20)
per = (nblocks + lanes - 1) // lanes
parts = [t[i * per * bs: min(t.numel(), (i + 1) * per * bs)] for i in range(lanes)]
outs = [None] * lanes
def work(i):
    with torch.cuda.stream(streams[i]):
        o, b = encs[i].encode(parts[i], opts=opts, blocks_only=True)
        outs[i] = (o, b)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(lanes)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); dt = time.time() - t0
    tot = sum(o[0].numel() for o in outs)
    print(f"{lanes} lanes x batch {batch_mib} MiB, {mib} MiB: {dt*1e3:.0f} ms = {mib*1.048576/dt:.0f} MB/s, ratio {tot/t.numel():.4f}", flush=True)
