#!/bin/bash
# A/B of library builds on C2 and C4, 15 timed frames each: tools/ab_libs.sh lib1.so lib2.so ... (names under raytracer_amd/; "default" = the product build)
for lib in "$@"; do
  if [ "$lib" != default ]; then export RT_AMD_LIB=$PWD/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 15 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s c2 %.3f ms   c4 %.3f ms' % ('$lib', d['ms_per_step'], d['also'][0]['ms_per_step']))" || exit 1
done
