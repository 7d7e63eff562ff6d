#!/bin/bash
# C5 frame time per library build, 6 frames: tools/ab_c5.sh lib1.so lib2.so ... (names under raytracer_amd/; "default" = the product build)
for lib in "$@"; do
  if [ "$lib" != default ]; then export RT_AMD_LIB=$PWD/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload c5 --also '' --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s c5 %.3f ms' % ('$lib', d['ms_per_step']))" || exit 1
done
