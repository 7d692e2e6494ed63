#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <mib> <preset>   (run on the GPU box; PMC passes only with --kernel-trace)
set -u
TAG=$1; MIB=$2; PRESET=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/prof_case.py $MIB $PRESET > $OUT/$name.log 2>&1
}
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<PY
import csv, glob, collections
for name in ("inst", "cyc", "fetch", "write"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "span_encode" in k or name in ("fetch", "write"):
            print(name, k, {c: int(v) for c, v in d.items()}, "dispatches", max(calls[(k, c)] for c in d))
PY
