#!/bin/bash
# Builds variants of the eight-phase GEMM harness: build_gemm8_bench.sh name "-DFLAGS" [name "-DFLAGS" ...]
cd "$(dirname "$0")" && mkdir -p bin
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-result -DNDEBUG $2 gemm8_bench.cpp -o bin/gemm8_bench_$1 2>&1 | grep -E "error" &
  shift 2
done
wait
ls bin/
