"""Reads a rocprofv3 kernel trace (CSV) of scripts/share_timeline.py: the last frame (from its last k_primary to the end)."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r.get("Queue_Id", "?")))
rows.sort()
n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_primary")]
first = starts[-n_lanes]
frame = rows[first:]
t0 = frame[0][0]
end = max(r[1] for r in frame)
print("last frame: %d launches, span %.3f ms" % (len(frame), (end - t0) / 1e6))
# union of busy intervals, and time with >= 2 kernels resident
events = sorted([(r[0], 1) for r in frame] + [(r[1], -1) for r in frame])
busy = multi = 0
depth, last = 0, t0
for t, d in events:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
print("some kernel resident %.3f ms, two or more %.3f ms, none %.3f ms" % (busy / 1e6, multi / 1e6, (end - t0 - busy) / 1e6))
by = {}
for s, e, name, q in frame:
    k = by.setdefault(name, [0, 0])
    k[0] += 1
    k[1] += e - s
for name, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s %3d launches %8.3f ms" % (name[:60], n, t / 1e6))
if len(sys.argv) > 3:
    for s, e, name, q in frame:
        print("%9.3f %9.3f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, name[:70]))
