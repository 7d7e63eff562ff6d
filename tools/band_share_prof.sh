#!/bin/bash
# per-kernel durations of one rank's share of the C4 tile split (N given): tools/band_share_prof.sh N
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/bs_$1
BS_MODE=serial timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bs_$1 -- python3 $R/tools/band_share.py $1 > $R/gpurun_out/bs_$1.log 2>&1
python3 - <<PY
import csv,glob
t=glob.glob("$R/gpurun_out/bs_$1/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(t)) if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last pass: from the last wf_gen on
last=max(i for i,r in enumerate(rows) if "wf_gen" in r["Kernel_Name"])
t0=int(rows[last]["Start_Timestamp"]); busy=0
for r in rows[last:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; busy+=d
    print("   %-30s start %7.3f  dur %7.3f ms" % (r["Kernel_Name"][:30], (int(r["Start_Timestamp"])-t0)/1e6, d))
print("   span %.3f ms, kernels busy %.3f ms" % ((int(rows[-1]["End_Timestamp"])-t0)/1e6, busy))
PY
