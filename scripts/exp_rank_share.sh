#!/bin/bash
# Per-rank share times (scripts/rank_times.py) with each experimental build under raytracer_amd/lib/exp/.
for so in raytracer_amd/lib/exp/librt_*.so; do
  echo "== $(basename $so .so)"
  RT_AMD_LIB=$PWD/$so timeout -k 10 200 python scripts/rank_times.py 2>/dev/null || exit 1
done
