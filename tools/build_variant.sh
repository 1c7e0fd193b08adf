#!/bin/bash
# Build an A/B copy of the library into ab_variants/<name>.so with extra -D flags.  Usage: tools/build_variant.sh name -DX=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../ray-tracer-challenge_amd"
mkdir -p ../ab_variants /tmp/abv/$name
for f in rtc_api.cpp rtc_scene.cpp rtc_loader.cpp rtc_kernels.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-function "$@" -c csrc/$f -o /tmp/abv/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../ab_variants/$name.so /tmp/abv/$name/*.o
echo built ab_variants/$name.so
