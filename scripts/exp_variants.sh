#!/bin/bash
# Runs bench.py on scene7@640x480 with each experimental build under raytracer_amd/lib/exp/.
for so in raytracer_amd/lib/exp/librt_*.so; do
  name=$(basename $so .so)
  RT_AMD_LIB=$PWD/$so timeout -k 10 120 python -u bench.py --workload scene7 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-16s step %.1f ms  closest %.1f TF (%.3f)  shadow %.1f TF (%.3f)  closest_ms %.1f shadow_ms %.1f' % ('$name', d['ms_per_step'], r['achieved'], r['frac'], r['shadow_kernel']['achieved'], r['shadow_kernel']['frac'], d['per_rank_max_ms']['closest_ms'], d['per_rank_max_ms']['shadow_ms']))
" || exit 1
done
