#!/usr/bin/env python3
"""tools/c3_swap.py [sets] -- which array makes a set of C3 arrays slow?  `sets` sets alive together; the fastest and the slowest are
found, then the step is timed on the fast set with ONE array at a time taken from the slow set, and the other way round."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["v", "y", "s", "a", "b", "row_off", "var_off", "result", "q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values"]


def main(sets=6, steps=10):
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
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    S = []
    for _ in range(sets):
        ins = [torch.from_numpy(h).to(dev) for h in host_in]
        roff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
        voff = torch.empty((chunk + 1,), dtype=torch.int64, device=dev)
        res = torch.empty((chunk, 2), dtype=torch.int64, device=dev)
        cols = pg.Columns.allocate(10 * chunk, 15 * chunk, dev)
        S.append(ins + [roff, voff, res] + [getattr(cols, n) for n in NAMES[8:]])

    def timed(arr):
        p = [t.data_ptr() for t in arr]
        cc = _lib.ColumnsC(*p[8:])

        def call():
            assert lib.pg_scalar_mix_planned_batch(eng._h, *p[:5], chunk, p[5], p[6], None, 3, 5, 0, C.byref(cc), p[7], sp) == 0
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

    base = [timed(s) for s in S]
    fast, slow = base.index(min(base)), base.index(max(base))
    print(json.dumps({"ms_per_step_by_set": base, "fast": fast, "slow": slow}), flush=True)
    for a, b, tag in ((fast, slow, "fast set, one array from the slow set"), (slow, fast, "slow set, one array from the fast set")):
        out = {}
        for k, n in enumerate(NAMES):
            arr = list(S[a])
            arr[k] = S[b][k]
            out[n] = timed(arr)
        print(json.dumps({tag: out}), flush=True)
    # groups
    for tag, idx in (("inputs", range(0, 5)), ("selectors", range(8, 13)), ("wires", range(13, 16)), ("var_values", [16])):
        arr = list(S[fast])
        for k in idx:
            arr[k] = S[slow][k]
        arr2 = list(S[slow])
        for k in idx:
            arr2[k] = S[fast][k]
        print(json.dumps({"group": tag, "fast_set_with_slow_group": timed(arr), "slow_set_with_fast_group": timed(arr2)}), flush=True)
    print(json.dumps({"ptrs_fast": {n: hex(t.data_ptr()) for n, t in zip(NAMES, S[fast])}, "ptrs_slow": {n: hex(t.data_ptr()) for n, t in zip(NAMES, S[slow])}}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
