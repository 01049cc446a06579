#!/bin/bash
# rocprofv3 passes over one bench.py command; run on the GPU box:  scripts/profile_r3.sh <tag> [bench arguments]
# Writes gpurun_out/prof_<tag>/{trace,pmc_sq,pmc_fetch,pmc_write}; scripts/summarize_profile.py condenses them for profiles/.
# One kernel-trace pass and three counter passes, each a run of its own.
set -u
TAG=${1:-r3}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-as-written $*"
echo "== kernel trace" && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
tail -1 "$OUT/trace.log" | cut -c1-300
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-as-written $*"
echo "== pmc sq" && timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o pmc -- python3 "$ROOT/bench.py" $PMCARGS > "$OUT/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
echo "== pmc fetch" && timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" $PMCARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
echo "== pmc write" && timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/bench.py" $PMCARGS > "$OUT/pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -size +20M -delete
du -sh "$OUT"
