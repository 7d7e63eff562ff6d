#!/bin/bash
# A/B of library builds: tools/abl_lib.sh "bench args" lib1.so lib2.so ... (names under raytracer_amd/; "" = the default build)
extra=$1; shift
for lib in "$@"; do
  if [ -n "$lib" ]; then export RT_AMD_LIB=$PWD/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  echo "== ${lib:-default}"
  bash tools/abl2.sh "$extra" 2>/dev/null || exit 1
done
