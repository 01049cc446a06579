#!/bin/bash
# Brute-force leg of the bench (roofline object): staged test (default) and the test as written (--plain-test).
#   scripts/r2_brute.sh [workload] [crop WxH]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WL=${1:-scene7_4k}; CROP=${2:-768x432}
for mode in "" "--plain-test"; do
    timeout -k 10 400 python bench.py --workload $WL --brute-crop $CROP $mode --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/tmp_bf.log 2>gpurun_out/tmp_bf.err || { tail -5 gpurun_out/tmp_bf.err; exit 1; }
    python3 - "$mode" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_bf.log').read().strip().splitlines()[-1])
r=d['roofline']; s=r['shadow_kernel']
print('%-12s %s frame %.1f ms | closest: %.1f ms/launch x%d = %.2f TFLOP/s (%.3f)  shadow: %.1f ms/launch x%d = %.2f TFLOP/s (%.3f) same image %s counters %s' % (sys.argv[1] or 'staged', d['config']['workload'], d['ms_per_step'], r['avg_launch_ms'], r['launches'], r['achieved'], r['frac'], s['avg_launch_ms'], s['launches'], s['achieved'], s['frac'], r['image_identical_to_default_mode'], r['counters_identical_to_default_mode']))
PY
done
