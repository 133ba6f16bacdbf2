#!/usr/bin/env python3
"""tools/values_pmc.py [tables] -- the workload of tools/values_pmc.sh: the witness refresh of C2's circuit (34.7 GB of assignments)
into `tables` tables alive together, CALLS launches per table in turn, TWO rounds; prints the launches' median ms per table
(events) so that the counters of a dispatch (rocprofv3 --pmc, collected around this very process) can be set beside the time of
the table it wrote: dispatch d of pg::emit_kernel<RangeCheckGD, EMIT_VALUES> wrote table (d // CALLS) % tables."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CALLS = 3


def main(tables=6):
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
    out = []
    for rnd in range(2):
        row = []
        for t in T:
            ms = []
            for _ in range(CALLS):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                assert lib.pg_range_check_values_batch(eng._h, C.byref(mn.c), C.byref(mx.c), wit.data_ptr(), chunk, t.data_ptr(), sp) == 0
                e1.record(stream)
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            row.append(round(sorted(ms)[CALLS // 2], 3))
        out.append(row)
    print(json.dumps({"calls_per_table": CALLS, "tables": tables, "ms_by_table_round0": out[0], "ms_by_table_round1": out[1],
                      "table_ptrs": [hex(t.data_ptr()) for t in T]}), flush=True)


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
