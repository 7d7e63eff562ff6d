"""Basic blocks of one kernel in a device-only assembly listing (see isa_mix.py), with their instruction counts:
   python tools/isa_blocks.py /tmp/rr.s wf_trace_kernel"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'\n(_Z\w*%s\w*):(.*?)\.Lfunc_end' % sys.argv[2], s, re.S)
body = m.group(2).split('\n')
blocks = []; cur = ['entry', []]
for l in body:
    t = l.strip()
    if not t or t.startswith((';', '//')): continue
    lab = re.match(r'^(\.LBB\d+_\d+):', t)
    if lab:
        blocks.append(cur); cur = [lab.group(1), []]; continue
    if t.startswith('.'): continue
    cur[1].append(t)
blocks.append(cur)
for name, ins in blocks:
    ops = [i.split()[0] for i in ins]
    v = sum(o.startswith('v_') for o in ops)
    tags = []
    def cnt(pred): return sum(1 for o in ops if pred(o))
    for label, pred in (('cvtub', lambda o: 'cvt_f32_ubyte' in o), ('f64', lambda o: '_f64' in o), ('ds', lambda o: o.startswith('ds_')),
                        ('gload', lambda o: o.startswith('global_load')), ('gstore', lambda o: o.startswith('global_store')),
                        ('atom', lambda o: o.startswith('global_atomic')), ('rcp', lambda o: 'rcp' in o or 'div_' in o),
                        ('scratch', lambda o: o.startswith('scratch')), ('pk', lambda o: o.startswith('v_pk_')),
                        ('cndmask', lambda o: o.startswith('v_cndmask')), ('readlane', lambda o: 'readlane' in o or 'readfirstlane' in o),
                        ('bperm', lambda o: 'bpermute' in o), ('salu', lambda o: o.startswith('s_') and not o.startswith(('s_waitcnt', 's_cbranch', 's_branch', 's_nop')))):
        c = cnt(pred)
        if c: tags.append(f'{label}{c}')
    br = [i.split()[-1] for i in ins if i.startswith(('s_cbranch', 's_branch'))]
    print(f"{name:11s} n={len(ins):4d} valu={v:4d} {' '.join(tags):70s} -> {' '.join(br)}")
