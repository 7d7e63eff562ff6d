"""Every counter of a tools/pmc.sh directory, per dispatch of the kernels whose name holds a given substring:
   python tools/pmc_dump.py DIR SUBSTRING   (dispatches in launch order, one line each)"""
import csv, glob, collections, sys
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + '/pass*/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if sys.argv[2] in k:
            per[(k, int(r['Dispatch_Id']))][r['Counter_Name']] += float(r['Counter_Value'])
    byk = collections.defaultdict(list)
    for (k, d), v in per.items():
        byk[k].append((d, v))
    for k, l in byk.items():
        l.sort()
        for i, (d, v) in enumerate(l):
            rows[(k, i)].update(v)
for (k, i), v in sorted(rows.items()):
    print(k[-28:], '#%d' % i, '  '.join('%s %.4g' % (n.replace('SQ_', '').replace('_sum', ''), x) for n, x in sorted(v.items())))
