#!/bin/bash
# The grown boxes of the tree against the listed-line tests (rt_commit.hip, tri_box_walk): the soak's collapsed-axis and sword
# styles (seed % 10 == 7 / 8 from 200 000 on) and the render-path test with (a) no listed-line test anywhere and no box growth,
# (b) no listed-line test but the grown boxes, (c) the product build.  Usage: tools/thin_teeth.sh FIRST COUNT
first=${1:-900000}; count=${2:-600}
for lib in diag_rt_diag_no_thin_all__2.so diag_rt_diag_no_thin_all.so default; do
  if [ "$lib" != default ]; then export RT_AMD_LIB=$PWD/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  echo "== $lib"
  timeout -k 10 400 python -u tests/soak_parity.py --first $first --count $count | grep -E "FAIL|seeds equal" | cut -c1-220 | tail -n 12
  timeout -k 10 200 python -u -m pytest tests/test_parity_gpu.py -q -k collinear --timeout 150 --timeout-method thread 2>&1 | tail -n 1
done
