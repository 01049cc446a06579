#!/usr/bin/env python3
"""Turns the rocprofv3 sqlite outputs of scripts/profile_r3.sh into the text summaries kept under profiles/.

    python scripts/summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [profiles/traffic.json]
"""
import glob
import json
import os
import sqlite3
import sys


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = ["%-10s %-14s %-14s %-8s %s" % ("calls", "total_ms", "avg_ms", "pct", "kernel")]
    for name, calls, total, avg, pct in rows:
        lines.append("%-10d %-14.3f %-14.4f %-8.3f %s" % (calls, total / 1e3, avg / 1e3, pct, name))
    return "\n".join(lines)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


def counters(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
                      "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    table = {}
    for kernel, counter, n, total, dur in rows:
        table.setdefault(short(kernel), {})[counter] = (n, total, dur)
    return table


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = []
    trace = glob.glob(os.path.join(src, "trace", "*.db"))
    if trace:
        out.append("== rocprofv3 --kernel-trace --stats  (durations in ms)\n" + kernel_stats(trace[0]))
    merged = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for path in glob.glob(os.path.join(src, sub, "*.db")):
            for kernel, cs in counters(path).items():
                merged.setdefault(kernel, {}).update(cs)
    if merged:
        out.append("\n== rocprofv3 --pmc, summed over the dispatches of each kernel (three separate passes)")
        for kernel, cs in merged.items():
            if not any(k in kernel for k in ("k_closest", "k_shadow", "k_primary", "k_shade", "k_resolve", "k_hit")):
                continue
            out.append("\n%s" % kernel)
            for counter, (n, total, dur) in sorted(cs.items()):
                out.append("  %-24s dispatches %-5d sum %-22.1f per dispatch %-18.1f" % (counter, n, total, total / n))
            g = lambda c: cs[c][1] if c in cs else None
            if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
                out.append("  derived: VALU instructions per wave            %.0f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
            if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
                out.append("  derived: ACTIVE_INST_VALU / WAVE_CYCLES          %.3f  (both in quad-cycles, per wave)" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
            if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
                # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles; 1024 SIMDs; GUI_ACTIVE is summed over 8 XCDs.
                # The two counters come from different passes of the same command.
                n_i, n_g = cs["SQ_INSTS_VALU"][0], cs["GRBM_GUI_ACTIVE"][0]
                cycles = g("GRBM_GUI_ACTIVE") / 8.0 * (float(n_i) / n_g)
                out.append("  derived: VALU issue utilisation = 2*INSTS_VALU / (kernel cycles * 1024 SIMDs) = %.3f" % (2.0 * g("SQ_INSTS_VALU") / (cycles * 1024.0)))
            if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
                out.append("  derived: WAIT_INST_ANY / WAVE_CYCLES             %.3f" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
            if g("FETCH_SIZE") is not None:
                n = cs["FETCH_SIZE"][0]
                out.append("  derived: HBM-side read bytes per dispatch        %.3e  (FETCH_SIZE is in KiB; x2 gfx950 correction for 16-B/lane streaming reads applied)" % (2.0 * 1024.0 * g("FETCH_SIZE") / n))
            if g("WRITE_SIZE") is not None:
                n = cs["WRITE_SIZE"][0]
                out.append("  derived: HBM-side write bytes per dispatch       %.3e  (WRITE_SIZE in KiB)" % (1024.0 * g("WRITE_SIZE") / n))
            if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None and cs["FETCH_SIZE"][2] and cs["WRITE_SIZE"][2]:
                # bytes over the kernel's own time in each pass (durations are in ns); the HBM3E peak is about 8000 GB/s
                rate = 2.0 * 1024.0 * g("FETCH_SIZE") / cs["FETCH_SIZE"][2] + 1024.0 * g("WRITE_SIZE") / cs["WRITE_SIZE"][2]
                out.append("  derived: HBM-side traffic                        %.1f GB/s = %.4f of 8000 GB/s" % (rate, rate / 8000.0))
            if g("SQ_LDS_IDX_ACTIVE") is not None and g("GRBM_GUI_ACTIVE"):
                # LDS_IDX_ACTIVE counts, per CU, the cycles its LDS is working on an instruction; 256 CUs; GUI_ACTIVE is summed over 8 XCDs
                out.append("  derived: LDS busy = LDS_IDX_ACTIVE / (kernel cycles * 256 CUs) = %.3f" % (g("SQ_LDS_IDX_ACTIVE") / (g("GRBM_GUI_ACTIVE") / 8.0 * 256.0)))
            if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
                out.append("  derived: L2 hit rate                             %.4f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
            if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
                out.append("  derived: LDS bank-conflict cycles / LDS active   %.4f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
            if g("GRBM_GUI_ACTIVE") is not None:
                n, total, dur = cs["GRBM_GUI_ACTIVE"]
                out.append("  derived: effective clock = GUI_ACTIVE/8/duration = %.3f GHz" % (total / 8.0 / dur))
    for log in ("trace.log", "pmc_sq.log", "brute_force.log", "synthetic.log"):
        path = os.path.join(src, log)
        if os.path.exists(path):
            for line in open(path):
                if line.startswith('{"metric"'):
                    out.append("\n== bench.py line of the %s run\n%s" % (log.replace(".log", ""), line.strip()))
    # Figures of the brute-force triangle scans that bench.py reports in its roofline object (traffic, VALU issue
    # utilisation), keyed by the workload of the PMC runs; and VALU instructions per (wave, triangle) step.
    line = None
    for log in ("pmc_sq.log", "trace.log"):
        path = os.path.join(src, log)
        for text in open(path) if os.path.exists(path) else []:
            if text.startswith('{"metric"'):
                line = json.loads(text)
    if line and "roofline" in line and len(sys.argv) > 3:
        plain = "brute<1,0>" in str(line["roofline"].get("kernel"))
        label = line["config"]["workload"] + ("/brute-force-plain" if plain else "/brute-force")
        rows = line["roofline"].get("kernels", {})
        entry = {}
        for tag, names, row, rays_per_lane in (("closest", ("k_closest_tri_brute",), "closest_triangle", 1 if plain else 2),
                                               ("shadow", ("k_shadow_tri_pretest", "k_shadow_tri_brute"), "shadow_triangle", 1),
                                               ("closest_ellipsoid", ("k_closest_ell_staged", "k_closest_ell<"), "closest_ellipsoid", 2)):
            cs = [v for k, v in merged.items() if any(n in k for n in names)]
            pairs = rows.get(row, {}).get("pairs")
            if not cs or not pairs:
                continue
            cs = cs[0]
            g = lambda c: cs[c][1] if c in cs else None
            if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
                entry[tag + "_hbm_bytes_per_launch"] = (2.0 * 1024.0 * g("FETCH_SIZE") + 1024.0 * g("WRITE_SIZE")) / cs["FETCH_SIZE"][0]
            if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
                cycles = g("GRBM_GUI_ACTIVE") / 8.0 * (float(cs["SQ_INSTS_VALU"][0]) / cs["GRBM_GUI_ACTIVE"][0])
                entry[tag + "_valu_issue_utilisation"] = 2.0 * g("SQ_INSTS_VALU") / (cycles * 1024.0)
            if g("SQ_INSTS_VALU"):
                entry[tag + "_valu_instructions_per_pair_and_wave"] = g("SQ_INSTS_VALU") / (pairs / 64.0)
        import datetime
        import subprocess
        try:
            commit = subprocess.run(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except OSError:
            commit = ""
        entry["source"] = "%s: rocprofv3 PMC passes of `bench.py` on this workload, run by the builder on %s%s" % (
            os.path.basename(dst), datetime.date.today().isoformat(), (" at commit " + commit) if commit else "")
        table = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
        table[label] = entry
        json.dump(table, open(sys.argv[3], "w"), indent=1, sort_keys=True)
        out.append("\n== brute-force scans (%s) -> %s\n%s" % (label, sys.argv[3], json.dumps(entry, indent=1)))
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out)[:6000])


if __name__ == "__main__":
    main()
