#!/bin/bash
# round 6, last session: compile-flag variants of the kernel library against each other on one 1368 MiB batch, and the
# instruction-cache counters of the parse kernel (76 KB of code against 64 KB of instruction cache per two CUs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_flags.txt; : > $O
export AB_SHA=1 AB_REPS=2
for v in base o3 os maxilp maxmem trk bias100 nounroll; do
  lib=xz_amd/libxz_amd_$v.so; [ "$v" = base ] && lib=xz_amd/libxz_amd.so
  [ -f $lib ] || continue
  XZ_AMD_LIB=$PWD/$lib python tools/gpu_ab.py 1368 6 $v: 2>&1 | grep -v amdgpu.ids >> $O
done
P=gpurun_out/pmc_icache; mkdir -p $P
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $P/a -o p --output-format csv -- python tools/prof_case.py 1368 6 > $P/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $P/b -o p --output-format csv -- python tools/prof_case.py 1368 6 > $P/b.log 2>&1
python3 - <<PY > gpurun_out/r06_pmc_icache.txt
import csv, glob, collections
for name in ("a", "b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$P/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in sorted(agg.items()):
        if any(t in k for t in ("k_parse", "k_find", "k_model", "k_rc")):
            print(name, k, {c: int(v) for c, v in d.items()})
PY
rm -rf $P/a $P/b
cat $O gpurun_out/r06_pmc_icache.txt
