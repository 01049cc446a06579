#!/bin/bash
# Full GPU suite, then one bench line per workload given (frame time, scan times, work per ray).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_suite.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
: > gpurun_out/bench_lines.log
for w in "$@"; do
    extra=""
    case $w in scene5|scene6) extra="--width 3840 --height 2160";; esac
    timeout -k 10 300 python bench.py --workload $w $extra --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/tmp_bench.log 2>&1 || { tail -5 gpurun_out/tmp_bench.log; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/tmp_bench.log').read().strip().splitlines()[-1])
r=d['rays_per_step']
print('%-22s %8.2f ms  %8.1f Mrays/s  closest %.1f shadow %.1f  child tests/ray %.0f leaf/ray %.1f' % (d['config']['workload'], d['ms_per_step'], d['value'], d['per_rank_max_ms']['closest_ms']/d['steps'], d['per_rank_max_ms']['shadow_ms']/d['steps'], sum(d['pair_pretests_per_step'])/r, sum(d['queued_pairs_per_step'])/r))" >> gpurun_out/bench_lines.log
done
cat gpurun_out/bench_lines.log
