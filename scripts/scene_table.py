#!/usr/bin/env python3
"""One bench.py line per configuration the reference is quoted on, condensed into a table.

    python scripts/scene_table.py [out.json]          (on a box with one MI355X; about ten minutes)

For each of the nine provided scenes at its own 640x480, for scene5 / scene6 / scene7 at 3840x2160 and for the synthetic scene:
the default mode's frame time and Mrays/s, the brute-force mode (RT_FLAG_NO_FILTER) on the whole frame (640x480) or a centred
crop (3840x2160): Mrays/s, ray-primitive tests/s per kind and scan with the share of the FP32 peak, and the unmodified
reference on this box's host cores: the whole frame at 640x480 (render section only), a centred crop at 3840x2160.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["scene1", "scene2", "scene3", "scene4-diffuse", "scene4-emission", "scene4-specular", "scene5", "scene6", "scene7"]
RUNS = [(name, [], ["--cpu-crop", "640x480", "--steps", "50", "--warmup", "5"]) for name in SMALL]
# the reference renders scene5 / scene6 at 3840x2160 in seconds: whole frames; scene7 takes hours: a centred crop
RUNS += [(name + "@4k", ["--width", "3840", "--height", "2160"], ["--steps", "20", "--warmup", "3"] + (["--cpu-crop", "3840x2160"] if name != "scene7" else [])) for name in ("scene5", "scene6", "scene7")]
RUNS += [("synthetic", [], ["--steps", "10", "--warmup", "2"])]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r3_scenes.json")
    table = {}
    for label, size, rest in RUNS:
        workload = label.split("@")[0]
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--no-as-written"] + size + rest
        t0 = time.time()
        done = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        if done.returncode != 0:
            print(label, "FAILED", done.stderr[-500:], flush=True)
            continue
        line = json.loads(done.stdout.strip().splitlines()[-1])
        table[label] = line
        r, b, c = line["roofline"], line["brute_force"], line["cpu_baseline"]
        kinds = "; ".join("%s %.3g tests/s %.1f%%" % (k, v["pairs_per_s"], 100 * v["frac"]) for k, v in r["kernels"].items())
        print("%-16s %8.3f ms %8.1f Mrays/s | brute force %8.2f Mrays/s, %.3g tests/s, %.1f%% of peak [%s] | CPU %s: %.2f s, %.4f Mrays/s (%s) | identical: %s %s  (%.0f s)"
              % (label, line["ms_per_step"], line["value"], b["mrays_s"], b["tests_per_s"], 100 * b["frac_of_fp32_peak"], kinds, c["kind"], c["seconds"], c["value"],
                 c["sample"].split(" of")[0], r["image_identical_to_default_mode"], c["image_identical_to_gpu"], time.time() - t0), flush=True)
        with open(out_path, "w") as f:
            json.dump(table, f, indent=1)




def markdown(path):
    """A Markdown table (one row per configuration) from the JSON this script wrote."""
    table = json.load(open(path))
    print("| configuration | frame, default mode | Mrays/s | brute force: Mrays/s | tests/s (all kinds) | share of FP32 peak: all kinds; per scan | CPU reference (host threads) | Mrays/s | images identical (brute force = default; CPU = GPU) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for label, line in table.items():
        r, b, c = line["roofline"], line["brute_force"], line["cpu_baseline"]
        short = {"closest_triangle": "tri", "closest_sphere": "sph", "closest_ellipsoid": "ell", "shadow_triangle": "tri↓", "shadow_sphere": "sph↓", "shadow_ellipsoid": "ell↓"}
        kinds = ", ".join("%s %.0f" % (short[k], 100 * v["frac"]) for k, v in r["kernels"].items())
        size = line["config"]["workload"].split("@")[1]
        whole = c["sample"].startswith("centred %s-pixel" % size)
        print("| %s | %.3f ms | %.0f | %.1f | %.3g | %.1f %%; %s | %.2f s, %s (%d) | %.3g | %s; %s |" % (
            line["config"]["workload"], line["ms_per_step"], line["value"], b["mrays_s"], b["tests_per_s"], 100 * b["frac_of_fp32_peak"], kinds,
            c["seconds"], "whole frame" if whole else c["sample"].split(" of")[0].replace("centred ", "").replace("-pixel crop", " crop"), c["threads_used"], c["value"],
            "yes" if r["image_identical_to_default_mode"] else "NO", "yes" if c["image_identical_to_gpu"] else "NO"))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--markdown":
        markdown(sys.argv[2])
    else:
        main()
