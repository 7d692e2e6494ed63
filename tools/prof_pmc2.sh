#!/bin/bash
# PMC passes over one single-batch encode (tools/prof_case.py): instruction mix and wave-cycle counters of every
# kernel, then HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes).  usage (GPU box): tools/prof_pmc2.sh <tag> <mib> <preset>
set -u
TAG=$1; MIB=$2; PRESET=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- python tools/prof_case.py $MIB $PRESET > $OUT/$name.log 2>&1; }
run inst SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for name in ("inst", "cyc", "fetch", "write"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k.replace("(anonymous namespace)::", "")[:48]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
    for k, d in sorted(agg.items()):
        if any(t in k for t in ("k_span", "k_find", "k_link", "k_sa_", "onesweep", "k_hash")):
            print(name, k, {c: int(v) for c, v in d.items()}, "dispatches", max(calls[(k, c)] for c in d))
PY
cat $OUT/summary.txt
rm -rf $OUT/inst $OUT/cyc $OUT/fetch $OUT/write
