#!/bin/bash
# bench.py at maxdepth 1 and 2 and the default: separates eye rays, first bounce and the rest (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for d in 1 2 -1; do
  timeout -k 10 200 python3 -u "$ROOT/bench.py" --no-cpu-baseline --maxdepth $d "$@" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('maxdepth %s: step %.1f ms closest %.1f shadow %.1f | rays %.3e tests %.3e | packet-kept pairs %s | queued %s | full %.3e' % (d['config']['maxdepth'], d['ms_per_step'], d['per_rank_max_ms']['closest_ms'], d['per_rank_max_ms']['shadow_ms'], d['rays_per_step'], d['tests_per_step'], ['%.3e' % v for v in d['pair_pretests_per_step']], ['%.3e' % v for v in d['queued_pairs_per_step']], d['full_tests_per_step']))
" || exit 1
done
