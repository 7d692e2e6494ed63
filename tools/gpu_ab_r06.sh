B="python bench.py --steps 2 --warmup 1 --no-ratio --no-extra-configs --no-cpu-baseline --no-host-to-host"
for cfg in "base:" "st2low:XZAMD_ST2_LOW=1" "b1100:XZAMD_BATCH_MIB=1100" "b1100low:XZAMD_BATCH_MIB=1100 XZAMD_ST2_LOW=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs $B 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', d['value'], d['ms_per_step'], d['stage_ms_last_step'])"
done
