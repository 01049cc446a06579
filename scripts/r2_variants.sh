#!/bin/bash
# Brute-force leg for each alternative build of the library given (raytracer_amd/lib/<name>.so), then the default build.
#   WL=synthetic CROP=768x432 scripts/r2_variants.sh variant_a variant_b
cd "$(dirname "$0")/.."
for v in "$@" librt_amd; do
    echo "== $v"
    RT_AMD_LIB=$PWD/raytracer_amd/lib/$v.so scripts/r2_brute.sh ${WL:-scene7_4k} ${CROP:-768x432} > gpurun_out/tmp_variant.log || exit 1
    sed -n 1p gpurun_out/tmp_variant.log
done
