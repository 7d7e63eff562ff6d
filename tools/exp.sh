#!/bin/bash
# quick ablation sweep on the GPU box: prints Mrays/s, ms/frame, rays/frame per variant
for v in "--depth 1" "--depth 2" "--depth 4" "--depth 4 --spp 4" "--depth 4 --spp 1" "--workload c1 --width 1920 --height 1080 --spp 16"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $v | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', '| Mrays/s %.0f  ms %.2f  rays/frame %.3g  nodes/ray %.1f tris/ray %.1f' % (d['value'], d['ms_per_step'], d['config']['rays_per_frame'], r['nodes_per_ray'], r['tris_per_ray']))"
done
