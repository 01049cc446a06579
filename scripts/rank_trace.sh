#!/bin/bash
# Kernel trace of one rank's share of the 4K scene7 frame:  scripts/rank_trace.sh WORLD RANK
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/rank_trace
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o t -- python3 "$ROOT/scripts/rank_trace.py" "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
tail -1 "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
tot = 0
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:12]:
    tot += total
    print("%6d calls %10.3f ms total %9.4f ms avg %6.2f%%  %s" % (calls, total / 1e3, avg / 1e3, pct, name.replace("(anonymous namespace)::", "").split("(")[0]))
print("sum of kernel time over 3 renders: %.3f ms" % (tot / 1e3))
PY
