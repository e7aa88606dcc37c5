#!/bin/bash
# Host-side AddressSanitizer check of the C ABI's launch planners (sanitizers on the CPU build only - GPU
# ASan / xnack builds are not available on the pool).  Builds api.cpp and the files that hold host-side planning code
# (dispatchers, split-K / unit plans, workspace queries, argument checks) with -fsanitize=address for the HOST half only and
# runs asan_driver.c, which calls every entry point that returns before launching a kernel (workspace queries over a grid of
# layer shapes incl. the benchmark's, the tuning table, every argument-check failure path).  No GPU needed.
set -e
cd "$(dirname "$0")"
mkdir -p build/asan
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -fsanitize=address -fno-omit-frame-pointer"
pids=()
for f in conv_igemm conv_pp conv_wgrad conv_wgrad_rows conv_wgrad_1x1; do
  if [ ! -f build/asan/$f.o ] || [ $f.hip -nt build/asan/$f.o ] || [ common.h -nt build/asan/$f.o ] || [ conv_common.h -nt build/asan/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/asan/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc $FLAGS -c api.cpp -o build/asan/api.o
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address build/asan/*.o -o build/asan/libsempyr_asan.so
hipcc -x c++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -I../../include asan_driver.c -o build/asan/asan_driver -Lbuild/asan -lsempyr_asan -Wl,-rpath,"$PWD/build/asan"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0 build/asan/asan_driver
