#!/usr/bin/env python3
"""tools/values_instances.py [tables] [rounds] -- the witness refresh of C2's circuit (34.7 GB of assignments) into `tables` tables
alive together, timed in turn `rounds` times: does the time belong to the table (where it lies) or to the moment?"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(tables=5, rounds=4):
    import numpy as np
    import torch
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib, synth
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    chunk = 1 << 20
    eng = pg.Engine(0)
    wit = torch.from_numpy(synth.random_scalars(chunk).view(np.int64)).to(dev)
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
    T = [torch.empty((chunk * 1034, 4), dtype=torch.int64, device=dev) for _ in range(tables)]
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    def timed(t, steps=5):
        def call():
            assert lib.pg_range_check_values_batch(eng._h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), chunk, t.data_ptr(), sp) == 0
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(steps):
            call()
        e1.record(stream)
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / steps, 3)

    t0 = time.perf_counter()
    for r in range(rounds):
        print(json.dumps({"round": r, "t_s": round(time.perf_counter() - t0, 2), "ms_by_table": [timed(t) for t in T]}), flush=True)
    print(json.dumps({"table_ptrs": [hex(t.data_ptr()) for t in T]}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
