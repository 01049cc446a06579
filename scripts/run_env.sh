#!/bin/bash
# bench lines under environment settings given as "VAR=V VAR2=V2" strings, one run each:  scripts/run_env.sh workload "settings"...
cd "$(dirname "$0")/.."
w=$1; shift
extra=""; case $w in scene5|scene6) extra="--width 3840 --height 2160";; esac
for rep in 1 2; do for setting in "$@"; do
  env $setting timeout -k 10 200 python bench.py --workload $w $extra --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-44s' % '$setting', d['config']['workload'], round(d['ms_per_step'],2), 'box tests/ray %.0f' % (sum(d['pair_pretests_per_step'])/d['rays_per_step']), d['frame_identical_to_single_rank_render'])" || exit 1
done; done
