"""tools/next_rows_allocated.py -- on the GPU box: the f-rows (pg_composer_permutation, pg_composer_materialize) on a 270 M-row composer built by
range_check_batch (every item allocates its witness: closed form, nothing foreign) against range_check_allocated_batch (the witnesses are
Variables allocated before: one foreign Variable per item -- the reference's own signature), in one process, twice.  Median of 7 calls, ms."""
import sys, time, json
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth
eng = pg.Engine(0)
S = pg.BlsScalar.from_int
batch = 1 << 18
wit = torch.from_numpy(synth.random_scalars(batch, seed=1).view(np.int64)).to("cuda:0")
for kind in (sys.argv[1:] or ["own", "allocated", "own", "allocated"]):
    if kind == "ragged":  # C4's shape: per-item public bounds (rows and Variables by prefix sums, wires read back)
        import bench
        mr, wt = bench.c4_inputs(1 << 19, seed=5)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
        dev = pg.StandardComposer(eng, 3 + (1 << 19) * 516 + 8, 5 + (1 << 19) * 520 + 8)
        dev.max_bound_ragged_batch(d(mr), d(wt))
    elif kind == "mix":  # C3's shape: ten rows and fifteen Variables per item
        import bench
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
        nmix = 1 << 23
        dev = pg.StandardComposer(eng, 3 + nmix * 10 + 8, 5 + nmix * 15 + 8)
        dev.scalar_mix_batch(*[d(x) for x in bench.mix_inputs(nmix, seed=6)])
    else:
        dev = pg.StandardComposer(eng, 3 + batch * 1031 + 8, 5 + batch * 1035 + 8)
    if kind in ("ragged", "mix"):
        pass
    elif kind == "own":
        dev.range_check_batch(S(0), S(2**254), wit)
    else:
        first = dev.add_input_batch(wit)
        vars_ = torch.arange(first, first + batch, dtype=torch.int64, device="cuda:0")
        dev.range_check_allocated_batch(S(0), S(2**254), vars_, wit)
    n = dev.circuit_size(); padded = 1 << (n - 1).bit_length()
    res = {}
    for name, fn in (("permutation", lambda: dev.permutation(padded)), ("materialize", dev.materialize)):
        ms = []
        for _ in range(7):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3); del r
        ms.sort(); res[name] = round(ms[3], 3)
    print(json.dumps({"kind": kind, "rows": n, **res}), flush=True)
    del dev; import gc; gc.collect(); torch.cuda.empty_cache()
