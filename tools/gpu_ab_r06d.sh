# A/B: literal coders of the model walks in LDS (XZAMD_WALK_LITG=0) or in global memory (=1)
B="--warmup 1 --no-ratio --no-extra-configs --no-cpu-baseline --no-host-to-host"
for litg in 0 1; do
  for sz in "4096 2" "512 4"; do
    set -- $sz
    XZAMD_WALK_LITG=$litg python bench.py --size-mib $1 --steps $2 $B 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('litg=$litg', $1, d['value'], d['ms_per_step'], d['stage_ms_last_step'])"
  done
done
