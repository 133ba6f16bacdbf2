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

    def slab_layout(slots_gib):
        """nine arrays in one slab at the given GiB offsets (q_m q_l q_r q_o q_c w_l w_r w_o var_values)"""
        sizes = [G * 32] * 5 + [G * 8] * 3 + [V * 32]
        al = 2 << 20
        total = int(max(o * GiB + n for o, n in zip(slots_gib, sizes))) + 2 * al
        slab = torch.empty((total // 8,), dtype=torch.int64, device=dev)
        first = ((-slab.data_ptr()) % al) // 8
        at = [first + int(o * GiB) // al * al // 8 for o in slots_gib]
        iv = sorted((a, a + n // 8) for a, n in zip(at, sizes))
        assert all(a1 <= b0 for (_, a1), (b0, _) in zip(iv, iv[1:])) and iv[-1][1] <= slab.numel()
        sel = [slab[at[c]:at[c] + G * 4].view(G, 4) for c in range(5)]
        wc = [slab[at[5 + c]:at[5 + c] + G] for c in range(3)]
        vv = slab[at[8]:at[8] + V * 4].view(V, 4)
        cols = pg.Columns(*sel, *wc, vv)
        cols.slab = slab
        return cols

    L = {
        "A separate": None,
        "B Columns.allocate(spread_gib=26)": "B",
        "selectors 24 apart, then wires 24 apart, var_values": [0, 24, 48, 72, 96, 120, 144, 168, 176],
        "selectors 24 apart, wires + var_values right after the last": [0, 24, 48, 72, 96, 97, 98, 99, 100],
        "selectors 16 apart, wires 16 apart after them": [0, 16, 32, 48, 64, 80, 96, 112, 120],
        "selectors 24 apart, wires + var_values before the first": [4, 28, 52, 76, 100, 0, 1, 2, 3],
    }
    out = {k: [] for k in L}
    for _ in range(trials):
        for name, spec in L.items():
            cols = pg.Columns.allocate(G, V, dev) if spec is None else (pg.Columns.allocate(G, V, dev, spread_gib=26) if spec == "B" else slab_layout(spec))
            out[name].append(timed(cols))
            del cols
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
