#!/bin/bash
# Brute-force bench lines: staged test (default) and the test as written (--plain-test), scene7 at a given size.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
W=${1:-1280}; H=${2:-720}
for mode in "" "--plain-test"; do
    timeout -k 10 300 python bench.py --brute-force $mode --width $W --height $H --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/tmp_bf.log 2>gpurun_out/tmp_bf.err || { tail -5 gpurun_out/tmp_bf.err; exit 1; }
    python3 - "$mode" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_bf.log').read().strip().splitlines()[-1])
r=d['roofline']
print('%-12s frame %.1f ms closest: %.1f ms/launch x%d = %.2f TFLOP/s (%.3f)  shadow: %.1f ms/launch = %.2f TFLOP/s (%.3f)' % (sys.argv[1] or 'staged', d['ms_per_step'], r['avg_launch_ms'], r['launches'], r['achieved'], r['frac'], r['shadow_kernel']['avg_launch_ms'], r['shadow_kernel']['achieved'], r['shadow_kernel']['frac']))
PY
done
