#!/usr/bin/env python3
"""SURVEY.md 8(f)1, measured: the stock `xz` binary of the image with LD_PRELOAD=libxz_amd_preload.so on a tmpfs file.
`xz` feeds lzma_code through 8 KiB buffers (src/xz/file_io.h:14-18, coder.c:1190-1300), so this number sits below
`host_to_host` of bench.py; the same command with XZ_AMD_DISABLE=1 (the real liblzma on the host cores) runs beside
it on a smaller file.  usage: tools/cli_bench.py [MiB=4096] [cpu_MiB=256] [preset=6]  -> one JSON line"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xz_amd  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cpu_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
preset = sys.argv[3] if len(sys.argv) > 3 else "6"
xz = shutil.which("xz")
pre = os.path.join(ROOT, "xz_amd", "libxz_amd_preload.so")
tmp = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (mib + cpu_mib + 64) << 20 else "/tmp"
src = os.path.join(tmp, "xzamd_cli_bench.bin")
xz_amd.corpus_text(mib << 20, seed=1000).tofile(src)


def run(path, env_extra, reps):
    best = None
    out_bytes = 0
    runs = []
    for _ in range(reps):
        env = dict(os.environ, LD_PRELOAD=pre, **env_extra)
        t0 = time.perf_counter()
        p = subprocess.run(f"{xz} -T0 -{preset} -c {path} | wc -c", shell=True, capture_output=True, env=env)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": p.stderr.decode()[-500:]}
        out_bytes = int(p.stdout.split()[0])
        runs.append(round(dt, 2))
        best = dt if best is None or dt < best else best
    n = os.path.getsize(path)
    return {"MB/s": round(n / best / 1e6, 1), "seconds": round(best, 2), "in_bytes": n, "out_bytes": out_bytes, "ratio": round(out_bytes / n, 5), "runs_s": runs}


res = {"tool": "xz " + subprocess.run([xz, "--version"], capture_output=True, text=True).stdout.split("\n")[0],
       "command": f"LD_PRELOAD=libxz_amd_preload.so xz -T0 -{preset} -c FILE | wc -c", "tmp": tmp,
       "gpu": run(src, {}, 3)}
small = os.path.join(tmp, "xzamd_cli_bench_small.bin")
with open(src, "rb") as f, open(small, "wb") as g:
    g.write(f.read(cpu_mib << 20))
res["cpu_same_binary_XZ_AMD_DISABLE"] = run(small, {"XZ_AMD_DISABLE": "1"}, 1)
os.remove(src)
os.remove(small)
print(json.dumps(res), flush=True)
