#!/bin/bash
# Profiles of the bench command for profiles/: kernel-trace stats + separate PMC passes (never combined with other
# trace domains).  usage (GPU box): bash tools/prof_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python - "$@" > $OUT/meta.json <<'PY'
import json, sys
a = sys.argv[1:]
def opt(name, default):
    return a[a.index(name) + 1] if name in a else default
print(json.dumps({"bench_args": a, "corpus": opt("--corpus", "text"), "preset": opt("--preset", "6"),
                  "size_mib": opt("--size-mib", "4096"), "bcj": "--bcj" in a}))
PY
B="--no-cpu-baseline --no-host-to-host --no-ratio"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o b --output-format csv -- python bench.py "$@" --steps 2 --warmup 1 $B > $OUT/bench_under_trace.json 2> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc/$C -o p --output-format csv -- python bench.py "$@" --steps 1 --warmup 0 $B > $OUT/bench_under_$C.json 2> $OUT/$C.log
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/pmc/SQ -o p --output-format csv -- python bench.py "$@" --steps 1 --warmup 0 $B > $OUT/bench_under_SQ.json 2> $OUT/SQ.log
# lane utilisation of the vector ALU and scalar-unit occupancy (one pass; counters the device does not offer fail the pass only)
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $OUT/pmc/SQ2 -o p --output-format csv -- python bench.py "$@" --steps 1 --warmup 0 $B > $OUT/bench_under_SQ2.json 2> $OUT/SQ2.log
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json $OUT/meta.json > $OUT/pmc_summary.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls $OUT
