#!/bin/bash
for v in "--workload c4" "--workload c3 --steps 2 --warmup 1" "--workload c5 --steps 2 --warmup 1"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $v | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$v', '| Mrays/s %.0f  ms %.2f  rays/frame %.3g  nodes/ray %.1f tris/ray %.1f slow %.2g build_ms %.1f upload_s %.1f' % (d['value'], d['ms_per_step'], c['rays_per_frame'], r['nodes_per_ray'], r['tris_per_ray'], r['slow_path_frac'], c['bvh_build_ms'], c['scene_upload_s']))"
done
