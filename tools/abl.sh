#!/bin/bash
# A/B of alternative builds: tools/abl.sh LIB... ; prints C2 and C4 frame times per library
for lib in "$@"; do
  for v in "--workload c2" "--workload c4" "--workload c5 --steps 2"; do
    RT_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $v | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib $v', '| Mrays/s %.0f  ms %.2f nodes/ray %.1f' % (d['value'], d['ms_per_step'], r['nodes_per_ray']))" || exit 1
  done
done
