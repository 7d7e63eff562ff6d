#!/bin/bash
for lib in librt_amd.so ablate_rt_ablate_math.so ablate_rt_ablate_rng.so; do
 for v in "--depth 4" "--depth 1"; do
  RT_AMD_LIB=$PWD/raytracer_amd/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $v | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib $v', '| Mrays/s %.0f  ms %.2f  rays/frame %.3g' % (d['value'], d['ms_per_step'], d['config']['rays_per_frame']))"
 done
done
