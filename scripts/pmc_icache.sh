#!/bin/bash
# One counter pass for the instruction cache over one bench.py command (GPU box):  scripts/pmc_icache.sh <tag> [bench arguments]
set -u
TAG=${1:-ic}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU -d "$OUT/pmc_ic" -o pmc -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-brute-force-leg "$@" > "$OUT/pmc_ic.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc_ic.log"; exit 1; }
python3 - "$OUT" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/pmc_ic/*.db")[0])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
t = {}
for k, c, n, v in rows:
    k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    t.setdefault(k, {})[c] = (n, v)
for k, c in sorted(t.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", (0, 0))[1]):
    g = lambda n: c.get(n, (0, 0.0))[1]
    if g("SQ_BUSY_CYCLES") < 1e6:
        continue
    print("%-32s x%-3d icache req %.3e hit %.3e miss %.3e (miss/req %.4f)  ifetch %.3e  valu %.3e  wait_any/wave_cycles %.3f  req per valu %.3f" % (
        k, c["SQ_BUSY_CYCLES"][0], g("SQC_ICACHE_REQ"), g("SQC_ICACHE_HITS"), g("SQC_ICACHE_MISSES"), g("SQC_ICACHE_MISSES") / max(g("SQC_ICACHE_REQ"), 1),
        g("SQ_IFETCH"), g("SQ_INSTS_VALU"), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQC_ICACHE_REQ") / max(g("SQ_INSTS_VALU"), 1)))
PY
