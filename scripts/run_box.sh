#!/bin/bash
# A/B of the box records: tests with every ray class on boxes, then bench lines per class mask.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rc=0


[ $rc -eq 0 ] || exit $rc
: > gpurun_out/box_bench.log
for mask in "$@"; do
    for w in scene7_4k synthetic; do
        RT_AMD_BOX=$mask timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/tmp_bench.log 2>&1 || exit 1
        python3 -c "
import json
d=json.loads(open('gpurun_out/tmp_bench.log').read().strip().splitlines()[-1])
r=d['rays_per_step']
print('box=%-3s %-22s %8.2f ms  closest %.1f shadow %.1f  child tests/ray %.0f leaf/ray %.1f' % ('$mask', d['config']['workload'], d['ms_per_step'], d['per_rank_max_ms']['closest_ms']/d['steps'], d['per_rank_max_ms']['shadow_ms']/d['steps'], sum(d['pair_pretests_per_step'])/r, sum(d['queued_pairs_per_step'])/r))" >> gpurun_out/box_bench.log
    done
done
cat gpurun_out/box_bench.log
