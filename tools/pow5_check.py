import sys; sys.path.insert(0,'.')
import numpy as np
from raytracer_amd import capi
ctx=capi.Context(0)
# floats in [0,1]: high 16 bits from 0x0000 to 0x3F80
hi=np.arange(0,0x3F81,dtype=np.uint32)
r=ctx.detmath(9, hi.view(np.float32))
bad5=(r.astype(np.int64)%65536); bad1000=(r.astype(np.int64)//65536)
print("pow5 mismatches", int(bad5.sum()), "of", len(hi)*65536, "; pow1000 shortcut mismatches", int(bad1000.sum()))
idx=np.nonzero(bad5)[0]; print(idx[:10], bad5[idx[:10]])
ctx.close()
