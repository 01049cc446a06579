#!/bin/bash
# Brute-force leg (staged test) of alternative builds of the library (raytracer_amd/lib/<name>.so) beside the default
# build, twice over so that drift of the box shows:  WL=synthetic CROP=512x288 scripts/ab_brute.sh variant_a [variant_b ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2; do
for v in "$@" librt_amd; do
    RT_AMD_LIB=$PWD/raytracer_amd/lib/$v.so timeout -k 10 300 python bench.py --workload ${WL:-synthetic} --brute-crop ${CROP:-512x288} --steps 1 --warmup 0 --no-cpu-baseline --no-as-written > gpurun_out/tmp_bf.log 2>gpurun_out/tmp_bf.err || { tail -5 gpurun_out/tmp_bf.err; exit 1; }
    python3 - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/tmp_bf.log').read().strip().splitlines()[-1])
r=d['roofline']; s=r['shadow_kernel']
print('%-16s %s | closest: %8.1f ms/launch = %6.2f TFLOP/s (%.3f)  shadow: %8.1f ms/launch = %6.2f TFLOP/s (%.3f, %.3e pairs) same image %s counters %s' % (sys.argv[1], d['config']['workload'], r['avg_launch_ms'], r['achieved'], r['frac'], s['avg_launch_ms'], s['achieved'], s['frac'], s['pairs'], r['image_identical_to_default_mode'], r['counters_identical_to_default_mode']))
PY
done
done
