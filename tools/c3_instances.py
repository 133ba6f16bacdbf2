#!/usr/bin/env python3
"""tools/c3_instances.py [sets] [rounds] [steps] -- placement or time?  `sets` complete sets of C3 arrays (inputs + outputs, 2.6 GB
each) alive TOGETHER in one process; the step is timed on every set in turn, `rounds` times over (steps calls back to back
each time).  A set that is slow every round is slow because of WHERE it lies; sets that speed up and slow down together follow
something that changes with time."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(sets=4, rounds=12, steps=10):
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
        S.append((ins, roff, voff, res, cols, cols.as_c()))

    def timed(k):
        ins, roff, voff, res, cols, cc = S[k]

        def call():
            assert lib.pg_scalar_mix_planned_batch(eng._h, *[t.data_ptr() for t in ins], chunk, roff.data_ptr(), voff.data_ptr(), None,
                                                   3, 5, 0, C.byref(cc), res.data_ptr(), sp) == 0
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(steps):
            call()
        e1.record(stream)
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / steps, 4)

    t0 = time.perf_counter()
    for r in range(rounds):
        print(json.dumps({"round": r, "t_s": round(time.perf_counter() - t0, 2), "ms_per_step_by_set": [timed(k) for k in range(sets)]}), flush=True)
        time.sleep(float(os.environ.get("ROUND_PAUSE_S", "0.5")))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
