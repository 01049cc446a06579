#!/bin/bash
# Kernel-time breakdown of one bench.py run (GPU box):  scripts/quick_trace.sh [bench args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/quick_trace
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:9]:
    print("%6d calls %10.1f ms total %9.2f ms avg %6.2f%%  %s" % (calls, total / 1e3, avg / 1e3, pct, name.replace("(anonymous namespace)::", "").split("(")[0]))
PY
grep '"metric"' "$OUT/log.txt" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.1f ms closest %.1f shadow %.1f' % (d['ms_per_step'], d['per_rank_max_ms']['closest_ms'], d['per_rank_max_ms']['shadow_ms']))"
