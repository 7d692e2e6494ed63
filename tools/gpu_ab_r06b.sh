# A/B: minimum encode span 512 KiB (product) vs 256 KiB (variant library), at the full 4 GiB job and at one rank's share of an 8-GPU run
B="--warmup 1 --no-ratio --no-extra-configs --no-cpu-baseline --no-host-to-host"
for lib in libxz_amd.so libxz_amd_es256.so; do
  for sz in "4096 2" "512 4" "1024 3"; do
    set -- $sz
    XZ_AMD_LIB=$PWD/xz_amd/$lib python bench.py --size-mib $1 --steps $2 $B 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', $1, d['value'], d['ms_per_step'], d['stage_ms_last_step'])"
  done
done
