cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, subprocess, time, shutil, sys
sys.path.insert(0, os.getcwd())
import xz_amd
src="/dev/shm/x.bin"; xz_amd.corpus_text(4096<<20, seed=1000).tofile(src)
pre=os.path.join(os.getcwd(),"xz_amd","libxz_amd_preload.so")
def run(envx, tag):
    env=dict(os.environ, LD_PRELOAD=pre, **envx)
    t0=time.perf_counter(); p=subprocess.run(f"xz -T0 -6 -c {src} | wc -c", shell=True, capture_output=True, env=env); dt=time.perf_counter()-t0
    print(tag, round(dt,2), p.stdout.split()[0].decode() if p.returncode==0 else p.stderr[-300:], flush=True)
for rep in range(2):
    run({}, "default")
    run({"XZAMD_BATCH_MIB":"648"}, "job648")
    run({"XZAMD_BATCH_MIB":"432"}, "job432")
run({"XZAMD_VERBOSE":"2"}, "default_verbose")
os.remove(src)
PY
