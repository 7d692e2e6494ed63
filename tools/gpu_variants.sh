#!/bin/bash
# A/B of library builds on the GPU box: tools/gpu_variants.sh <MiB> <preset> name[:ENV=V,...] ...  (name "base" = the product library)
MIB=$1; P=$2; shift 2
for v in "$@"; do
  name=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  lib=xz_amd/libxz_amd_$name.so; [ "$name" = base ] && lib=xz_amd/libxz_amd.so
  echo "== $v"
  env XZ_AMD_LIB=$PWD/$lib $(echo $envs | tr ',' ' ') python tools/gpu_timing_run.py $MIB $P 2>&1 | grep -v amdgpu.ids
done
