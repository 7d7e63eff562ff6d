#!/bin/bash
# careful A/B of library builds (10 timed steps, 3 warm-up, two rounds): tools/ab10.sh lib1.so lib2.so ... ("" = default build)
for round in 1 2; do
 for lib in "$@"; do
  if [ -n "$lib" ]; then export RT_AMD_LIB=$PWD/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  for v in c2 c4 c5; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --workload $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('round $round ${lib:-default} $v  %.3f ms  %.0f Mrays/s' % (d['ms_per_step'], d['value']))" || exit 1
  done
 done
done
