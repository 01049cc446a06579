#!/bin/bash
# Timeline of the kernel dispatches of one bench.py frame (GPU box): start offset, duration, grid, kernel.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/timeline
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 1 "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last k_primary pair backwards
prim = [i for i, r in enumerate(rows) if "k_primary" in r["Kernel_Name"]]
first = prim[-2]
t0 = int(rows[first]["Start_Timestamp"])
out = open(sys.argv[1] + "/timeline.txt", "w")
for r in rows[first:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    line = "%9.3f ms  +%8.3f ms  q%-3s grid %9d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?"), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)), name)
    out.write(line + "\n")
print(open(sys.argv[1] + "/timeline.txt").read())
PY
