#!/bin/bash
# Frame time of the small reference scenes at 3840x2160 for several values of RT_AMD_BRUTE_BELOW.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/small_scenes.log
for below in "$@"; do
    for w in scene1 scene2 scene3 scene4-specular scene6; do
        RT_AMD_BRUTE_BELOW=$below timeout -k 10 200 python bench.py --workload $w --width 3840 --height 2160 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/tmp_bench.log 2>&1 || exit 1
        python3 -c "
import json,sys
d=json.loads(open('gpurun_out/tmp_bench.log').read().strip().splitlines()[-1])
print('below=%s %-16s %8.2f ms  closest %.1f shadow %.1f' % ('$below', d['config']['workload'], d['ms_per_step'], d['per_rank_max_ms']['closest_ms']/d['steps'], d['per_rank_max_ms']['shadow_ms']/d['steps']))" >> gpurun_out/small_scenes.log
    done
done
cat gpurun_out/small_scenes.log
