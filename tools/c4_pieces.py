#!/usr/bin/env python3
"""tools/c4_pieces.py LIB -- the C4 step (pg_max_bound_ragged_plan_async + pg_max_bound_ragged_batch) through library LIB, a few
times; run under `rocprofv3 --kernel-trace` (tools/c4_timeline.sh) to see when the inversion pre-pass ends beside the emitter."""
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
mr_np, wt_np = bench.c4_inputs(chunk)
mr = torch.from_numpy(mr_np.view(np.int64)).to(dev)
wt = torch.from_numpy(wt_np.view(np.int64)).to(dev)
nb = torch.empty((chunk,), dtype=torch.int32, device=dev)
roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
res = torch.empty((chunk,), dtype=torch.int64, device=dev)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
h = C.c_void_p()
assert lib.pg_engine_create(0, C.byref(h)) == 0
lay = _lib.LayoutC()
assert lib.pg_max_bound_ragged_plan(h, mr.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), C.byref(lay), sp) == 0
cols = pg.Columns.allocate(int(lay.n_gates), int(lay.n_vars), dev)
cc = cols.as_c()
for _ in range(4):
    assert lib.pg_max_bound_ragged_plan_async(h, mr.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), sp) == 0
    assert lib.pg_max_bound_ragged_batch(h, mr.data_ptr(), wt.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), 3, 5,
                                         C.byref(cc), res.data_ptr(), sp) == 0
    torch.cuda.synchronize()
print("done")
