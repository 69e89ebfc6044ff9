import sys, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from compression_algorithms_amd import lz
from compression_algorithms_amd.context import Context
import test_fuzz_gpu as F
ctx=Context(0)
for kind in ("runs","lowent","text"):
    try:
        base=F._family(kind, 7, 20_000_000)
    except Exception as e:
        print(kind, "n/a", e); continue
    x=torch.from_numpy(base).cuda()
    p=lz.params('deflate')
    st=lz.compress(x,p,ctx); torch.cuda.synchronize()
    t0=time.perf_counter(); st=lz.compress(x,p,ctx); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(kind, round(x.numel()/dt/1e9,3), 'GB/s')
