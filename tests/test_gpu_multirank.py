"""N > 1 product path on the GPU box: two processes, each with its own Encoder context, shard the Blocks,
run Encoder.encode(blocks_only=True) on the device and gather the encoded Blocks to rank 0
(xz_amd.parallel.gather_stream).  A 1-GPU box has no second device for RCCL, so both ranks use GPU 0 and
the exchange runs over gloo on host tensors; what is exercised is the real encode + shard + gather +
framing code with world_size 2 (the N-GPU RCCL launch is bench.py --gpus N)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

import _oracle as o

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 1 << 18


def _worker(rank, world, port, data, preset, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import xz_amd
    from xz_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    enc = xz_amd.Encoder(0)
    nblocks = (len(data) + BS - 1) // BS
    lo, hi = parallel.shard_blocks(nblocks, rank, world)
    shard = data[lo * BS:hi * BS]
    opts = xz_amd.preset_options(preset)
    if shard:
        t = torch.frombuffer(bytearray(shard), dtype=torch.uint8).cuda()
        out, binfo = enc.encode(t, opts=opts, block_size=BS, blocks_only=True)
        blocks = out.cpu()
    else:
        blocks, binfo = torch.empty(0, dtype=torch.uint8), []
    stream = parallel.gather_stream(blocks, binfo, check=xz_amd.CHECK_CRC64)
    if rank == 0:
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        whole, _ = enc.encode(t, opts=opts, block_size=BS)
        q.put((stream.numpy().tobytes(), whole.cpu().numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("preset,nbytes", [(6, 7 * BS + 777), (1, 3 * BS), (6, BS // 3)])
def test_encode_blocks_only_and_gather_world2(preset, nbytes):
    data = o.corpus_mixed(nbytes, 21)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 2000) + preset
    procs = [ctx.Process(target=_worker, args=(r, 2, port, data, preset, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, whole = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == whole, "gathered Stream differs from the single-context Stream"
    r, dec, nb = o.orc_xz_decode(got, len(data) + 16)
    assert r == 0 and dec == data and nb == (nbytes + BS - 1) // BS
    if o.have_ref():
        rr, rdec = o.ref_decode(got, len(data) + 16)
        assert rr == 1 and rdec == data


def test_bench_two_ranks_oversubscribed():
    """`bench.py --gpus 2` end to end on this 1-GPU box (XZAMD_BENCH_OVERSUBSCRIBE=1: both ranks on GPU 0, gather over gloo):
    the launch path the driver uses for its N-GPU runs -- torch.distributed.run, one process per rank, Blocks sharded, the
    gather inside the timed region, max over ranks, ONE JSON line from rank 0."""
    import json
    import subprocess
    env = dict(os.environ, XZAMD_BENCH_OVERSUBSCRIBE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size-mib", "256", "--steps", "1",
                        "--warmup", "1", "--no-ratio", "--no-cpu-baseline", "--no-host-to-host"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["other_scaling"]["scaling"] == "weak" and d["other_scaling"]["value"] > 0
    assert "roofline" in d and d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo"
    # round 6: all ranks cut their Blocks out of ONE corpus, so the 2-rank Stream must be the 1-rank Stream of the same job
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size-mib", "256", "--steps", "1", "--warmup", "0", "--no-ratio",
           "--no-cpu-baseline", "--no-host-to-host", "--no-extra-configs", "--stream-sha"]
    shas = {}
    for g in (1, 2):
        q = subprocess.run(cmd + ["--gpus", str(g)], capture_output=True, text=True, timeout=900, env=env)
        assert q.returncode == 0, q.stderr[-2000:]
        dd = json.loads([l for l in q.stdout.splitlines() if l.startswith("{")][-1])
        shas[g] = (dd["stream_sha256"], dd["stream_bytes"])
    assert shas[1] == shas[2], shas
