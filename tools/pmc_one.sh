#!/bin/bash
# one counter pass of one workload per library build, the launches of one kernel (KERNEL=substring, default wf_trace_kernel): tools/pmc_one.sh "COUNTERS" "bench args" lib1.so lib2.so ... ("default" = the product build)
ctrs=$1; extra=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  if [ "$lib" != default ]; then export RT_AMD_LIB=$R/raytracer_amd/$lib; else unset RT_AMD_LIB; fi
  out=$R/gpurun_out/pmc1_$(echo $lib | tr -c 'A-Za-z0-9_\n' '_')
  rm -rf $out; mkdir -p $out
  timeout -s KILL 180 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass0 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also '' --inflight 1 $extra > $out/pass0.log 2>&1 || { echo "pass failed"; tail -3 $out/pass0.log; exit 1; }
  echo "== $lib"
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/pass0/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
rows=list(csv.DictReader(open(f)))
for r in rows:
    k=r["Kernel_Name"][:40]
    acc[(k,r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
byk=collections.defaultdict(list)
for (k,d),v in acc.items(): byk[k].append((int(d),v))
for k,l in byk.items():
    if "${KERNEL:-wf_trace_kernel}" not in k: continue
    l.sort()
    for d,v in l[-3:]:
        print("  ", k, d, " ".join("%s %.4g"%(a,b) for a,b in sorted(v.items())))
PY
done
