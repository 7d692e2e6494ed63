# A/B: waves per SIMD of the parse kernel (4 = product: 128 VGPRs, 120 spilled; 3: 168 VGPRs, 70 spilled; 2: 256 VGPRs)
B="--warmup 1 --no-ratio --no-extra-configs --no-cpu-baseline --no-host-to-host --size-mib 1368 --steps 3"
for lib in libxz_amd.so libxz_amd_w3.so libxz_amd_w2.so; do
  XZ_AMD_LIB=$PWD/xz_amd/$lib python bench.py $B 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', d['value'], d['ms_per_step'], d['stage_ms_last_step'])"
done
