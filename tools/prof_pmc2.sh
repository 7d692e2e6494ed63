#!/bin/bash
# usage: tools/prof_pmc2.sh <tag> <mib> <preset> [passes...]   (GPU box; PMC passes only with --kernel-trace)
# Each pass = "name:COUNTER,COUNTER,..."; a pass whose counters are rejected is skipped.
set -u
TAG=$1; MIB=$2; PRESET=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  rocprofv3 --kernel-trace --pmc ${ctrs//,/ } -d $OUT/$name -o p --output-format csv -- python tools/prof_case.py $MIB $PRESET > $OUT/$name.log 2>&1 || echo "pass $name failed"
done
python tools/pmc_summary.py $OUT $OUT/summary.json
