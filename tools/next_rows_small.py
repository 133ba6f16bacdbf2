"""tools/next_rows_small.py -- on the GPU box: the f-rows (pg_composer_permutation, pg_composer_materialize) on composers filled by ONE kind of
small batched append each (the scalar gadgets on existing Variables, the gate batches, the fused mix with failing items): median of 7 calls, ms,
and GB/s of what the call writes (sigma: 32 B per padded row; materialize: 328 B per row)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth
import bench
eng = pg.Engine(0)
S = pg.BlsScalar.from_int
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to("cuda:0")
log2 = int(os.environ.get("LOG2", "22"))
batch = 1 << log2
rng = np.random.default_rng(1)
for kind in (sys.argv[1:] or ["is_non_zero", "is_non_zero_failing", "select_zero", "select_one", "maybe_equal", "add", "rows", "mix_failing"]):
    comp = pg.StandardComposer(eng, 3 + 11 * batch, 5 + 16 * batch)
    scal = synth.random_scalars(batch, seed=3)
    if kind.endswith("failing"):
        scal[::101] = 0
    tv = lambda: torch.from_numpy(rng.integers(5, 5 + batch, size=batch).astype(np.int64)).to("cuda:0")
    if kind.startswith("mix"):
        v, y, s, a, b = bench.mix_inputs(batch, seed=6)
        v[::101] = 0
        comp.scalar_mix_batch(*[d(x) for x in (v, y, s, a, b)])
    else:
        comp.add_input_batch(d(scal))
        if kind.startswith("is_non_zero"):
            comp.is_non_zero_batch(tv())
        elif kind == "select_zero":
            comp.conditionally_select_zero_batch(tv(), tv())
        elif kind == "select_one":
            comp.conditionally_select_one_batch(tv(), tv())
        elif kind == "maybe_equal":
            comp.maybe_equal_batch(tv(), tv())
        elif kind == "add":
            comp.add_batch(S(3), tv(), S(5), tv(), S(7))
        else:
            comp.boolean_gate_batch(tv())
    n = comp.circuit_size(); padded = 1 << (n - 1).bit_length()
    res = {}
    for name, fn, nbytes in (("permutation", lambda: comp.permutation(padded), 32 * padded), ("materialize", comp.materialize, 328 * n)):
        ms = []
        for _ in range(7):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3); del r
        ms.sort(); res[name] = {"ms": round(ms[3], 3), "GBps": round(nbytes / ms[3] / 1e6)}
    print(json.dumps({"kind": kind, "rows": n, "padded": padded, **res}), flush=True)
    comp.close(); del comp
    import gc; gc.collect(); torch.cuda.empty_cache()
