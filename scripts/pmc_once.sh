#!/bin/bash
# One rocprofv3 PMC pass (SQ counters) over a bench.py run; prints the counters of the scan kernels:  scripts/pmc_once.sh [bench args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_once
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM -d "$OUT" -o pmc -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/*.db")[0])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
t = {}
for k, c, n, v in rows:
    k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    t.setdefault(k, {})[c] = (n, v)
for k, cs in t.items():
    if "closest" in k or "shadow" in k:
        print(k, {c: "%.3g" % (v / n) for c, (n, v) in cs.items()})
PY
