#!/bin/bash
# A/B of bench variants: tools/abl2.sh "args1" "args2" ... ; each on C2, C4, C5
for extra in "$@"; do
  for v in "--workload c2" "--workload c4" "--workload c5 --steps 2"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $v $extra | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$extra $v', '| Mrays/s %.0f  ms %.2f nodes/ray %.1f tris/ray %.1f' % (d['value'], d['ms_per_step'], r['nodes_per_ray'], r['tris_per_ray']))" || exit 1
  done
done
