#!/bin/bash
# Build libstreamvln_hip.so for gfx950 (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libstreamvln_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p build
pids=()
for f in gemm gemv attention misc preprocess decode_layer engine; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ ../../include/streamvln_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/gemm.o build/gemv.o build/attention.o build/misc.o build/preprocess.o build/decode_layer.o build/engine.o -o $OUT
echo "built $OUT"
