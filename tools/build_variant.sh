#!/bin/bash
# Build a measurement variant of the library next to the product: xz_amd/libxz_amd_<name>.so (select it with XZ_AMD_LIB).
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../xz_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -fno-strict-aliasing $FLAGS -c lzma_kernels.hip -o /tmp/lk_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libxz_amd_$NAME.so /tmp/lk_$NAME.o lzma_decode.o xzamd_host.o xzamd_stream.o xzamd_decode.o corpus.o \
  -Wl,-Bsymbolic -Wl,--version-script=libxz_amd.map -Wl,-soname,libxz_amd.so -lpthread
ls -la ../libxz_amd_$NAME.so
