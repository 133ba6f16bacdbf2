#!/usr/bin/env python3
"""uniform vs ragged max_bound at the same ladder length, accepted vs rejected witnesses (`reject` argument):
where does the C4 gap to C2 come from?  (1 GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth

dev = torch.device("cuda", 0)
eng = pg.Engine(0)
batch = 1 << 19
REJECT = len(sys.argv) > 1 and sys.argv[1] == "reject"
w_np = synth.random_scalars(batch)
if not REJECT:
    # accepted set: canonical values below 2^164, far under the 200-bit bound (random field elements are all rejected)
    w_np = np.tile(synth.scalars_from_ints([int(x) << 100 for x in synth.splitmix64(4096, 3)]), (batch // 4096, 1))
wit = torch.from_numpy(np.ascontiguousarray(w_np).view(np.int64)).to(dev)
mx = pg.BlsScalar.from_int(2**200 + 5)
lay = eng.max_bound_layout(mx, batch)
cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
nbytes = lay.n_gates * 184 + lay.n_vars * 32
stream = torch.cuda.current_stream(dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


t = timeit(lambda: eng.max_bound_batch(mx, wit, 3, 5, out=cols))
print("uniform max_bound n=%d: %.3f ms, %.0f GB/s" % (lay.num_bits, t, nbytes / t / 1e6))
# the same bound for every item through the ragged entry points
mr = torch.from_numpy(np.repeat(synth.scalars_from_ints([2**200 + 5]), batch, axis=0).view(np.int64)).to(dev)
import ctypes as C
from plonk_gadgets_amd import _lib
nb = torch.empty((batch,), dtype=torch.int32, device=dev)
roff = torch.empty((batch + 1,), dtype=torch.int64, device=dev)
voff = torch.empty((batch + 1,), dtype=torch.int64, device=dev)
res = torch.empty((batch,), dtype=torch.int64, device=dev)
lc = _lib.LayoutC()
assert eng._lib.pg_max_bound_ragged_plan(eng._h, mr.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), C.byref(lc), eng._stream()) == 0
assert int(lc.n_gates) == lay.n_gates
cc = cols.as_c()
t = timeit(lambda: eng._lib.pg_max_bound_ragged_batch(eng._h, mr.data_ptr(), wit.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), 3, 5, C.byref(cc), res.data_ptr(), eng._stream()))
print("ragged  max_bound (same bound): %.3f ms, %.0f GB/s" % (t, nbytes / t / 1e6))
