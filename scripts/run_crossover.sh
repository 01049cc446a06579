#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/crossover.log
for kind in triangles ellipsoids spheres; do
    for below in 0 1000000000; do
        RT_AMD_BRUTE_BELOW=$below timeout -k 10 300 python scripts/crossover.py $kind 16 64 128 256 512 1024 2048 4096 >> gpurun_out/crossover.log 2>&1
    done
done
