#!/bin/bash
# HBM traffic of the 360-node span kernel (nice_len > 128) on one 1920 MiB batch of x86-BCJ'd ELF at 9e:
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (with --kernel-trace only).  usage (GPU box): bash tools/prof_c5_pmc.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r03_c5
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  AB_REPS=1 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc/$C -o p --output-format csv -- python tools/gpu_ab.py 1920 0x80000009 corpus=elf c5:XZAMD_BCJ=1 > $OUT/$C.log 2>&1
  tail -1 $OUT/$C.log
done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summary.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
head -5 $OUT/pmc_summary.txt | cut -c1-300
