#!/usr/bin/env python3
"""tools/placement_policy.py [trials] -- three ways to allocate the C3 step's output arrays, `trials` times each in one process:
A separate allocations one after the other (what Columns.allocate did before round 4); B one slab, the arrays at equal strides
(Columns.allocate(spread_gib=...)); C separate allocations with a temporary spacer allocation before each, all spacers freed before
the step runs.  ms per step (10 steps back to back, median of 3)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GiB = 1 << 30


def main(trials=4, steps=10):
    import numpy as np
    import torch
    import bench
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    chunk = 1 << 20
    eng = pg.Engine(0)
    ins = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev) for x in bench.mix_inputs(chunk)]
    roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    G, V = 10 * chunk, 15 * chunk

    def timed(cols):
        cc = cols.as_c()

        def call():
            assert lib.pg_scalar_mix_planned_batch(eng._h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None,
                                                   3, 5, 0, C.byref(cc), res.data_ptr(), sp) == 0
        call()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(steps):
                call()
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / steps)
        return round(sorted(ts)[1], 4)

    def policy_c(spacer_gib):
        spacers, arrs = [], []
        shapes = [((G, 4),)] * 5 + [((G,),)] * 3 + [((V, 4),)]
        for (shape,) in shapes:
            spacers.append(torch.empty((spacer_gib * GiB,), dtype=torch.uint8, device=dev))
            arrs.append(torch.empty(shape, dtype=torch.int64, device=dev))
        del spacers
        torch.cuda.empty_cache()
        return pg.Columns(*arrs)

    out = {"A separate": [], "B slab 20 GiB stride": [], "B slab 26 GiB stride": [], "B slab 32 GiB stride": [], "B slab 44 GiB stride": []}
    for _ in range(trials):
        for name in out:
            if name.startswith("A"):
                cols = pg.Columns.allocate(G, V, dev)
            elif name.startswith("B"):
                cols = pg.Columns.allocate(G, V, dev, spread_gib=int(name.split()[2]))
            else:
                cols = policy_c(24 if "24" in name else 12)
            out[name].append(timed(cols))
            del cols
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
