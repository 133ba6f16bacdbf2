#!/usr/bin/env python3
"""tools/c3_pieces.py LIB -- the C3 step (pg_scalar_mix_planned_batch) through library LIB, a few times; run under
`rocprofv3 --kernel-trace --stats` (tools/c3_timeline.sh) to read every kernel's start, end and duration: the step's
launches run one after the other on one stream, so each has the chip to itself."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from plonk_gadgets_amd import _lib
import plonk_gadgets_amd as pg
import bench

lib = C.CDLL(sys.argv[1])
for fn, (r, a) in _lib.SIGNATURES.items():
    if hasattr(lib, fn):
        f = getattr(lib, fn)
        f.restype, f.argtypes = r, a
dev = torch.device("cuda", 0)
chunk = 1 << 20
ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(chunk)]
roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
cols = pg.Columns.allocate(10 * chunk, 15 * chunk, dev)
cc = cols.as_c()
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
h = C.c_void_p()
assert lib.pg_engine_create(0, C.byref(h)) == 0
two_step = len(sys.argv) > 2 and sys.argv[2] == "twostep"  # the plan as a call of its own, then pg_scalar_mix_batch (no look-back in the arithmetic launch)
for _ in range(5):
    if two_step:
        assert lib.pg_scalar_mix_plan_async(h, ins[0].data_ptr(), chunk, roff.data_ptr(), voff.data_ptr(), None, sp) == 0
        assert lib.pg_scalar_mix_batch(h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), 3, 5, 0, C.byref(cc),
                                       res.data_ptr(), sp) == 0
    else:
        assert lib.pg_scalar_mix_planned_batch(h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None, 3, 5, 0,
                                               C.byref(cc), res.data_ptr(), sp) == 0
    torch.cuda.synchronize()
print("done")
