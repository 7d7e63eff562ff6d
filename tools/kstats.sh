#!/bin/bash
# per-kernel durations of a bench run: tools/kstats.sh TAG [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/ks_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ks_$tag/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("$tag", r["Name"][:45], r["Calls"], "avg_ms %.3f" % (float(r["AverageNs"])/1e6))
PY
