#!/bin/bash
# frame time, box tests per ray and scene set-up time of the given bench workloads
cd "$(dirname "$0")/.."
for w in "$@"; do
  extra=""; case $w in scene5|scene6) extra="--width 3840 --height 2160";; esac
  timeout -k 10 200 python bench.py --workload $w $extra --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['ms_per_step'],2), 'box tests/ray %.0f' % (sum(d['pair_pretests_per_step'])/d['rays_per_step']), 'create ms %.0f' % d['setup_ms']['scene_create'])" || exit 1
done
