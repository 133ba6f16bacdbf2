import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth
eng = pg.Engine(0)
mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
for batch in (256, 1000, 4000, 16000, 65536):
    wit = torch.from_numpy(synth.random_scalars(batch, 3).view(np.int64)).to("cuda:0")
    lay = eng.range_check_layout(mn, mx, batch)
    cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0")
    res = torch.empty((batch,), dtype=torch.int64, device="cuda:0")
    for _ in range(2):
        eng.range_check_batch(mn, mx, wit, 3, 5, out=cols, result_vars=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.range_check_batch(mn, mx, wit, 3, 5, out=cols, result_vars=res)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(batch, "%.3f ms" % ms, "%.0f GB/s" % (batch * 222792 / ms / 1e6), flush=True)
