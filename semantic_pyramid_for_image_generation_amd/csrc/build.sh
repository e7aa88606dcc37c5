#!/bin/bash
# Builds libsempyr.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libsempyr.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
mkdir -p build
pids=()
for f in conv_igemm conv_pp fp8 conv_wgrad conv_wgrad_rows conv_wgrad_1x1 spectral_norm linear eltwise norm resample attention losses optim; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ conv_common.h -nt build/$f.o ] || [ ../../include/sempyr.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc $FLAGS -c api.cpp -o build/api.o
OBJS=""; for f in conv_igemm conv_pp fp8 conv_wgrad conv_wgrad_rows conv_wgrad_1x1 spectral_norm linear eltwise norm resample attention losses optim api; do OBJS="$OBJS build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $(realpath $OUT)"
