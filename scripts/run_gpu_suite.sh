#!/bin/bash
# Full GPU test suite, then bench lines for the scenes named on the command line (default: none).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/gpu_suite.log 2>&1
for w in "$@"; do
    timeout -k 10 200 python bench.py --workload "$w" --no-cpu-baseline --steps 5 --warmup 2 > "gpurun_out/bench_$w.log" 2>&1
done
