"""Static instruction mix per kernel of a device-only assembly listing:
   hipcc <flags of csrc/Makefile> --cuda-device-only -S -o /tmp/rr.s raytracer_amd/csrc/rt_render.hip
   python tools/isa_mix.py /tmp/rr.s"""
import re, sys
s = open(sys.argv[1]).read()
parts = re.split(r'\n(_Z[^\n:]*):', s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    lines = [l.strip() for l in body.split('\n')]
    lines = [l for l in lines if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]
    op = [l.split()[0] for l in lines]
    v = sum(o.startswith('v_') for o in op)
    f64 = sum('_f64' in o for o in op)
    ld = sum(o.startswith(('global_load', 'buffer_load', 'flat_load')) for o in op)
    st = sum(o.startswith(('global_store', 'buffer_store', 'flat_store')) for o in op)
    scr = sum(o.startswith('scratch_') for o in op)
    print(f"{name[:56]:56s} insts {len(op):6d} valu {v:6d} f64 {f64:5d} loads {ld:4d} stores {st:4d} scratch {scr:4d} waitcnt {sum(o == 's_waitcnt' for o in op)}")
