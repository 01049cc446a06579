#!/bin/bash
# Default-mode frame time of alternative builds of the library (raytracer_amd/lib/<name>.so) beside the default build,
# alternating so that clock drift of the box shows:  scripts/ab_default.sh variant_a [variant_b ...]
cd "$(dirname "$0")/.."
for round in 1 2; do
for v in "$@" librt_amd; do
    for wl in "synthetic --steps 4" "scene7 --width 3840 --height 2160 --steps 20"; do
        RT_AMD_LIB=$PWD/raytracer_amd/lib/$v.so timeout -k 10 200 python bench.py --workload $wl --no-brute-force-leg --no-cpu-baseline > gpurun_out/tmp_ab.json 2>gpurun_out/tmp_ab.err || exit 1
        python - "$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tmp_ab.json").read().strip().splitlines()[-1])
print("%-22s %-24s %9.3f ms  %8.1f Mrays/s  %7.1f executed tests per ray" % (sys.argv[1], d["config"]["workload"], d["ms_per_step"], d["value"], d["culled_mode"]["executed_tests_per_ray"]))
PY
    done
done
done
