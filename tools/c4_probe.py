import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth
dev = torch.device("cuda", 0); eng = pg.Engine(0); stream = torch.cuda.current_stream(dev)
batch = 1 << 19
mr_np, wt_np = bench.c4_inputs(batch)
def run(name, mr_np, wt_np):
    mr = torch.from_numpy(np.ascontiguousarray(mr_np).view(np.int64)).to(dev); wt = torch.from_numpy(np.ascontiguousarray(wt_np).view(np.int64)).to(dev)
    nb, roff, voff = eng.ragged_buffers(batch)
    lay = eng.max_bound_ragged_plan(mr, nb, roff, voff)
    cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev); res = torch.empty((batch,), dtype=torch.int64, device=dev)
    nbytes = lay.n_gates*184 + lay.n_vars*32
    ts=[]
    for i in range(6):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(stream); eng.max_bound_ragged_emit(mr, wt, nb, roff, voff, cols, res, 3, 5); e1.record(stream); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    t=sorted(ts)[len(ts)//2]; print(name, "%.3f ms"%t, "%.0f GB/s"%(nbytes/t/1e6), "rows/item %.1f"%(lay.n_gates/batch), flush=True)
    del cols; torch.cuda.empty_cache()
run("c4 as benched        ", mr_np, wt_np)
run("c4 witnesses all zero", mr_np, np.zeros_like(wt_np))
one = synth.scalars_from_ints([2**253 - 1])
run("one 253-bit bound, w=0", np.repeat(one, batch, axis=0), np.zeros_like(wt_np))
run("one 253-bit bound, c4 witnesses", np.repeat(one, batch, axis=0), wt_np)
