#!/bin/bash
# per-kernel durations of a bench run: tools/kstats.sh TAG [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --also '' "$@" > $R/gpurun_out/ks_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ks_$tag/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("$tag", r["Name"][:50], r["Calls"], "avg_ms %.3f total_ms %.2f" % (float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
t=glob.glob("$R/gpurun_out/ks_$tag/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(t)) if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame: from the last wf_gen launch of the first chunk on (frames = wf_gen launches / chunks per frame; one chunk assumed
# unless the launches say otherwise: a frame of several chunks shows its lanes' launches interleaved)
gens=[i for i,r in enumerate(rows) if "wf_gen" in r["Kernel_Name"]]
frames=7 if len(gens)%7==0 else max(1,len(gens))
n=len(rows)//frames
t0=int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    print("   %-28s q%-3s start %7.3f  dur %7.3f ms" % (r["Kernel_Name"][:28], r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
if rows[-n:]:
    print("   frame span %.3f ms" % ((int(rows[-1]["End_Timestamp"])-int(rows[-n]["Start_Timestamp"]))/1e6))
PY
