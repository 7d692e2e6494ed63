#!/bin/bash
# round 6 (a variant build: xzamd_encode_device_ halved max_blocks for a two-phase job that fits one batch and holds >= 1.9
# rounds of parse pieces, unless XZAMD_NO_SPLIT=1): one batch against two for 1 GiB and 512 MiB of the bench text = a rank's
# share of the metric's 4 GiB at 4 and 8 GPUs.  Same Stream, 1,954 -> 2,259 ms: the seed pieces (one wavefront per Block, 230 ms
# whatever the number of Blocks) and the partial iteration are paid per batch.  Not taken; profiles/r06_ab_split.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_split.txt; : > $O
B="--steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host --no-ratio --no-extra-configs --stream-sha"
for mib in 1024 512; do
  for v in 1 0; do
    echo "== $mib MiB XZAMD_NO_SPLIT=$v" >> $O
    if [ $v = 1 ]; then export XZAMD_NO_SPLIT=1; else unset XZAMD_NO_SPLIT; fi
    python bench.py --size-mib $mib $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stream_sha256'][:16], d['stage_ms_last_step'])" >> $O
  done
done
cat $O
