#!/bin/bash
# Build an A/B copy of the library into ab_variants/<name>.so with extra -D flags.  Usage: tools/build_variant.sh name -DX=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../ray-tracer-challenge_amd"
mkdir -p ../ab_variants /tmp/abv/$name; rm -f /tmp/abv/$name/*.o
FLAGS="-O3 -std=c++17 -fPIC -fno-fast-math --offload-arch=gfx950 -Wno-unused-function"
for f in rtc_api.cpp rtc_scene.cpp rtc_loader.cpp rtc_kernels.hip; do
  /opt/rocm/bin/hipcc $FLAGS -ffp-contract=off "$@" -c csrc/$f -o /tmp/abv/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../ab_variants/$name.so /tmp/abv/$name/*.o
echo built ab_variants/$name.so
