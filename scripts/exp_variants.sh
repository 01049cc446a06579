#!/bin/bash
# Runs bench.py (4K scene7, 1 step) with each experimental build under raytracer_amd/lib/exp/.
for so in raytracer_amd/lib/exp/librt_*.so; do
  name=$(basename $so .so)
  RT_AMD_LIB=$PWD/$so timeout -k 10 120 python -u bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-16s step %.1f ms  closest_ms %.1f shadow_ms %.1f' % ('$name', d['ms_per_step'], d['per_rank_max_ms']['closest_ms'], d['per_rank_max_ms']['shadow_ms']))
" || exit 1
done
