#!/usr/bin/env python3
"""tools/placement_span.py [sets] [spacer_GiB] -- the C3 step on sets of arrays spread over the card's whole memory: `sets` sets
with a spacer allocation (never touched) before each, so that the sets lie ~spacer_GiB apart in the order the driver hands memory
out.  Then the step on MIXED sets: selector column c from set (c * stride) % sets.  If memory far apart is different memory (ranks of
the 12-high stacks, partitions), a mixed set has more of it under its five lock-step streams."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["v", "y", "s", "a", "b", "row_off", "var_off", "result", "q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values"]


def main(sets=14, spacer_gib=16, steps=10):
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
    S, spacers = [], []
    for _ in range(sets):
        spacers.append(torch.empty((spacer_gib << 30,), dtype=torch.uint8, device=dev))
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

    print(json.dumps({"ms_per_step_by_set": [timed(s) for s in S]}), flush=True)
    for stride in (1, 2, 3, 5):
        out = []
        for base in range(0, sets, max(1, sets // 6)):
            arr = list(S[base])
            for c in range(5):
                arr[8 + c] = S[(base + c * stride) % sets][8 + c]
            out.append(timed(arr))
        print(json.dumps({"selector_columns_from_sets_stride": stride, "ms_per_step": out}), flush=True)
    # everything spread: array k of the call from set (k * 3) % sets
    out = []
    for base in range(0, sets, max(1, sets // 6)):
        out.append(timed([S[(base + 3 * k) % sets][k] for k in range(len(NAMES))]))
    print(json.dumps({"every_array_from_another_set": out}), flush=True)


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
