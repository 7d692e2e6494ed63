#!/bin/bash
# HBM traffic of a bench configuration for `roofline.traffic`: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (with
# --kernel-trace only) over ONE step of the bench command, summarised per dispatch (tools/pmc_summary.py) with the `_meta`
# bench.py matches a summary to a configuration by.  usage (GPU box): bash tools/prof_traffic.sh <tag> [bench args...]
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
B="--no-cpu-baseline --no-host-to-host --no-ratio --no-extra-configs"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc/$C -o p --output-format csv -- python bench.py "$@" --steps 1 --warmup 0 $B > $OUT/bench_under_$C.json 2> $OUT/$C.log
done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json $OUT/meta.json > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
rm -rf $OUT/pmc
ls $OUT
