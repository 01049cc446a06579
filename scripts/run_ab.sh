#!/bin/bash
# A/B of one environment switch on bench workloads:  scripts/run_ab.sh VAR workload...
cd "$(dirname "$0")/.."
var=$1; shift
for rep in 1 2; do for m in off on; do for w in "$@"; do
  extra=""; case $w in scene5|scene6) extra="--width 3840 --height 2160";; esac
  if [ $m = on ]; then export $var=${AB_VALUE:-1}; else unset $var; fi
  timeout -k 10 200 python bench.py --workload $w $extra --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$var $m', d['config']['workload'], round(d['ms_per_step'],2), d['frame_identical_to_single_rank_render'])" || exit 1
done; done; done
