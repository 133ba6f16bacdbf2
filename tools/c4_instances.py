#!/usr/bin/env python3
"""tools/c4_instances.py [instances] [steps] [slab] -- the C4 step (2^20 ragged max_bound items, 115.8 GB) on output arrays
allocated anew `instances` times in one process: ms per step of every instance.  slab = 1: the nine arrays carved out of ONE
allocation instead of nine; slab = N > 1: spread evenly over a slab of N GiB; slab = -S: Columns.allocate(spread_gib=S)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(instances=6, steps=8, slab=0, slab2=None):
    import numpy as np
    import torch
    import bench
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    chunk = 1 << 20
    eng = pg.Engine(0)
    mr_np, wt_np = bench.c4_inputs(chunk)
    mr = torch.from_numpy(mr_np.view(np.int64)).to(dev)
    wt = torch.from_numpy(wt_np.view(np.int64)).to(dev)
    nb = torch.empty((chunk,), dtype=torch.int32, device=dev)
    roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((chunk,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    lay = _lib.LayoutC()
    assert lib.pg_max_bound_ragged_plan(eng._h, mr.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), C.byref(lay), sp) == 0
    G, V = int(lay.n_gates), int(lay.n_vars)
    modes = [slab] if slab2 is None else [slab, slab2]  # two modes: alternating
    for inst in range(instances):
        slab = modes[inst % len(modes)]
        if slab < 0:  # Columns.allocate(spread_gib=-slab): selector columns that far apart, the rest behind the last
            cols = pg.Columns.allocate(G, V, dev, spread_gib=-slab)
            cc = cols.as_c()
            keep = cols
        elif slab:
            sizes = [G * 32] * 5 + [G * 8] * 3 + [V * 32]
            al = 2 << 20
            sizes_al = [(x + al - 1) // al * al for x in sizes]
            total = sum(sizes_al) if slab == 1 else slab << 30  # slab > 1: a slab of that many GiB, the arrays spread evenly over it
            gap = (total - sum(sizes_al)) // 8 // al * al
            # order: the five lock-step selector streams alternate with the wires and the variable table
            order = [0, 5, 1, 6, 2, 7, 3, 8, 4] if slab > 1 else list(range(9))
            buf = torch.empty((total,), dtype=torch.uint8, device=dev)
            ptrs, off = [0] * 9, (-buf.data_ptr()) % al
            for k in order:
                ptrs[k] = buf.data_ptr() + off
                off += sizes_al[k] + gap
            cc = _lib.ColumnsC(*ptrs)
            keep = buf
        else:
            cols = pg.Columns.allocate(G, V, dev)
            cc = cols.as_c()
            keep = cols

        def call():
            assert lib.pg_max_bound_ragged_batch(eng._h, mr.data_ptr(), wt.data_ptr(), chunk, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(),
                                                 3, 5, C.byref(cc), res.data_ptr(), sp) == 0
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(steps):
            call()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        print(json.dumps({"instance": inst, "slab": slab, "ms_per_step": round(ms, 3), "frac_of_8TBps": round((G * 184 + V * 32) / ms / 1e6 / 8000, 4),
                          "q_m": hex(cc.q_m or 0), "vars": hex(cc.var_values or 0)}), flush=True)
        del keep, cc
        buf = cols = None
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
