#!/bin/bash
# a library build with extra macros, beside the product's: tools/mklib.sh name.so -DRT_X=1 -DRT_Y=2 ... (-> raytracer_amd/name.so, for RT_AMD_LIB; never loaded by the product or the tests)
out=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -DRT_BUILD_ID=\"$(echo "$out $*" | tr -c 'A-Za-z0-9_=.\n' '_')\" "$@" \
  -shared -o ../$out rt_api.hip rt_commit.hip rt_render.hip -lz
