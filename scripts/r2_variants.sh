#!/bin/bash
# Brute-force bench line for each alternative build of the library given (raytracer_amd/lib/<name>.so), then the default build.
cd "$(dirname "$0")/.."
W=${W:-1280}; H=${H:-720}
for v in "$@" librt_amd; do
    echo "== $v"
    RT_AMD_LIB=$PWD/raytracer_amd/lib/$v.so scripts/r2_brute.sh $W $H | head -1
done
