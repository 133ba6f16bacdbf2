#!/usr/bin/env python3
"""tools/placement_sweep.py [slab_GiB] -- the C3 step against the DISTANCE between its arrays.  One slab; first the five selector
columns at spacing D (everything else in allocations of its own), then all seventeen arrays of the call at spacing D.  Separate torch
allocations put the columns 322 MiB apart (their size + 2 MiB)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MiB = 1 << 20
GiB = 1 << 30


def main(slab_gib=200, steps=10):
    import numpy as np
    import torch
    import bench
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    chunk = 1 << 20
    eng = pg.Engine(0)
    host_in = [np.ascontiguousarray(x).view(np.int64) for x in bench.mix_inputs(chunk)]
    ins = [torch.from_numpy(h).to(dev) for h in host_in]
    roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
    res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
    cols = pg.Columns.allocate(10 * chunk, 15 * chunk, dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    size = 10 * chunk * 32

    def timed(in_ptrs, off_ptrs, res_ptr, col_ptrs):
        cc = _lib.ColumnsC(*col_ptrs)

        def call():
            assert lib.pg_scalar_mix_planned_batch(eng._h, *in_ptrs, chunk, off_ptrs[0], off_ptrs[1], None, 3, 5, 0, C.byref(cc), res_ptr, sp) == 0
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

    own_in = [t.data_ptr() for t in ins]
    own_off = [roff.data_ptr(), voff.data_ptr()]
    own_cols = [getattr(cols, n).data_ptr() for n in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")]
    print(json.dumps({"separate_allocations": timed(own_in, own_off, res.data_ptr(), own_cols)}), flush=True)
    slab = torch.empty((slab_gib * GiB,), dtype=torch.uint8, device=dev)
    base = slab.data_ptr() + (-slab.data_ptr()) % (2 * MiB)
    out = {}
    for d in (() if os.environ.get("LAYOUTS_ONLY") else (0.33, 0.5, 1, 1.5, 2, 3, 4, 6, 8, 12, 16, 24, 32, 40)):
        D = int(d * GiB) // (2 * MiB) * (2 * MiB)
        if 4 * D + size > (slab_gib - 1) * GiB:
            break
        out["%g GiB" % d] = timed(own_in, own_off, res.data_ptr(), [base + c * D for c in range(5)] + own_cols[5:])
    print(json.dumps({"five_selector_columns_at_spacing": out}), flush=True)
    # all seventeen arrays in the slab (inputs copied in)
    out = {}
    sizes = [chunk * 32] * 5 + [(chunk + 1) * 8] * 2 + [chunk * 16] + [size] * 5 + [10 * chunk * 8] * 3 + [15 * chunk * 32]
    for d in (() if os.environ.get("LAYOUTS_ONLY") else (0.5, 1, 2, 3, 4, 6, 8, 11)):
        D = int(d * GiB) // (2 * MiB) * (2 * MiB)
        if 16 * D + sizes[-1] > (slab_gib - 1) * GiB or D < max(sizes):
            continue
        p = [base + k * D for k in range(17)]
        for k in range(5):
            slab[p[k] - slab.data_ptr():p[k] - slab.data_ptr() + sizes[k]].copy_(ins[k].view(torch.uint8).reshape(-1))
        out["%g GiB" % d] = timed(p[:5], p[5:7], p[7], p[8:])
    print(json.dumps({"all_seventeen_arrays_at_spacing": out}), flush=True)
    # which arrays need the distance?  slots of 8 GiB in the slab; a layout is a slot per array (inputs v y s a b | row_off var_off
    # result | q_m q_l q_r q_o q_c | w_l w_r w_o | var_values)
    def layout(slots):
        p = [base + int(sl * 8 * GiB) // (2 * MiB) * (2 * MiB) for sl in slots]
        iv = sorted((a, a + n) for a, n in zip(p, sizes))  # (an overlap would let one array's stores rewrite the prefix sums)
        assert all(a1 <= b0 for (_, a1), (b0, _) in zip(iv, iv[1:])) and iv[-1][1] <= slab.data_ptr() + slab.numel(), "layout overlaps"
        for k in range(5):
            slab[p[k] - slab.data_ptr():p[k] - slab.data_ptr() + sizes[k]].copy_(ins[k].view(torch.uint8).reshape(-1))
        return timed(p[:5], p[5:7], p[7], p[8:])
    near = lambda s0, n, step=0.4: [s0 + step * i for i in range(n)]  # (arrays 3.2 GiB apart: "together")
    L = {
        "everything together": near(0, 17),
        "selectors 32 GiB apart, rest together": near(0, 8) + [4, 8, 12, 16, 20] + near(4.5, 3) + [6],
        "selectors 32 apart, wires 32 apart": near(0, 8) + [4, 8, 12, 16, 20] + [3.2, 6.2, 10.2] + [14],
        "selectors + wires + var_values apart, inputs together": near(0, 8) + [3, 6, 9, 12, 15] + [18, 21, 23] + [22],
        "inputs apart too": [1, 4, 7, 10, 13] + near(0, 3, 0.1) + [3, 6, 9, 12, 15] + [18, 21, 23] + [22],
        "inputs 32 apart, outputs together": [0, 4, 8, 12, 16] + near(20, 12),
        "thirds: columns round robin over slots 0, 11, 22": near(0, 8, 0.05) + [0.5, 11, 22, 1, 11.5] + [22.5, 1.5, 12] + [23],
        "selectors apart, inputs apart, rest together": [1, 5, 9, 13, 17] + near(0, 3, 0.1) + [4, 8, 12, 16, 20] + near(21, 3) + [23],
    }
    out = {k: layout(v) for k, v in L.items()}
    print(json.dumps({"layouts_ms_per_step": out}), flush=True)
    # packed back to back in the slab (2-MiB aligned): what one allocation for everything gives
    p, off = [], 0
    for s in sizes:
        p.append(base + off)
        off += (s + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
    for k in range(5):
        slab[p[k] - slab.data_ptr():p[k] - slab.data_ptr() + sizes[k]].copy_(ins[k].view(torch.uint8).reshape(-1))
    print(json.dumps({"all_seventeen_packed": timed(p[:5], p[5:7], p[7], p[8:])}), flush=True)


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
