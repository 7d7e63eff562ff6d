#!/bin/bash
# last-frame launch timeline of C2 (or other bench args) per library build: tools/lvl.sh "bench args" lib1.so lib2.so ... ("" = default)
extra=$1; shift
for lib in "$@"; do
  if [ -n "$lib" ]; then export RT_AMD_LIB=$PWD/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  tag=lvl_$(echo "${lib:-default}_$extra" | tr -c 'A-Za-z0-9_\n' '_')
  rm -rf gpurun_out/ks_$tag
  echo "== ${lib:-default} $extra"
  bash tools/kstats.sh $tag $extra | grep -E "start|span" | awk '{printf "%s %s | ", substr($1,5,18), $(NF-1)} END {print ""}'
done
