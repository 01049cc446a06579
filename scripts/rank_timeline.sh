#!/bin/bash
# Timeline of the kernel dispatches of one rank's share of the 4K scene7 frame (GPU box):  scripts/rank_timeline.sh WORLD RANK
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/rank_timeline
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 "$ROOT/scripts/rank_trace.py" "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
tail -1 "$OUT/log.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prim = [i for i, r in enumerate(rows) if "k_primary" in r["Kernel_Name"]]
first = prim[-2] if len(prim) > 1 and int(rows[prim[-1]]["Start_Timestamp"]) - int(rows[prim[-2]]["Start_Timestamp"]) < 2_000_000 else prim[-1]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print("%9.3f ms  +%8.3f ms  q%-3s grid %9d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?"), int(r["Grid_Size_X"]), name))
PY
