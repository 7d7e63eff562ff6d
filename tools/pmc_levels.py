"""Per-dispatch figures of the trace / shade kernels from a tools/pmc.sh or tools/pmc0.sh directory:
VALU instructions, lanes active per VALU instruction, and (when the GRBM pass is there) issue-slot use."""
import csv, glob, collections, sys
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + '/pass*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'wf_shade' in k or 'wf_trace' in k:
            per[(k, int(r['Dispatch_Id']))][r['Counter_Name']] += float(r['Counter_Value'])
    byk = collections.defaultdict(list)
    for (k, d), v in per.items():
        byk[k].append((d, v))
    for k, l in byk.items():
        l.sort()
        for i, (d, v) in enumerate(l):
            rows[(k, i)].update(v)
for (k, i), v in sorted(rows.items()):
    g = v.get('GRBM_GUI_ACTIVE', 0) / 8; iv = v.get('SQ_INSTS_VALU', 0); av = v.get('SQ_ACTIVE_INST_VALU', 0); tc = v.get('SQ_THREAD_CYCLES_VALU', 0)
    print('%-30s #%2d  ms %.3f  VALU insts %.3g  issue %.2f  lanes %.1f' % (k[4:], i, g / 2.4e6, iv, iv * 4 / (1024 * g) if g else 0, tc / av if av else 0))
