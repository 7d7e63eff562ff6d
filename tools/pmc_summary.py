"""Sums rocprofv3 --pmc CSVs per kernel and counter (average per dispatch)."""
import csv, glob, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        nd[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {}
for k in acc:
    if not ("path_kernel" in k or "render_kernel" in k or "wf_" in k):
        continue
    out[k] = {c: acc[k][c] / max(1, len(nd[k][c])) for c in sorted(acc[k])}
    out[k]["dispatches"] = max(len(v) for v in nd[k].values())
json.dump(out, open(root + "/summary.json", "w"), indent=1)
for k, v in out.items():
    print(k)
    for c, x in v.items():
        print(f"   {c:34s} {x:16.4g}")
