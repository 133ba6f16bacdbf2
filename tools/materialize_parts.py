"""tools/materialize_parts.py [LOG2_BATCH] -- pg_composer_materialize with subsets of its output columns (any pointer may be
NULL): what the constant fills, the index reads and the assignment gathers each cost on a composer of 2^k x range_check(0, 2^254)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import _lib, synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
if len(sys.argv) > 2:  # another build of the library (tools/variants/lib_X.so)
    _lib.LIB_PATH = sys.argv[2]
eng = pg.Engine(0)
S = pg.BlsScalar.from_int
batch = 1 << lg
dev = pg.StandardComposer(eng, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
wit = torch.from_numpy(synth.uniform_below(batch, 2**254 + 2**250, seed=1).view(np.int64)).to("cuda:0")
dev.range_check_batch(S(0), S(2**254), wit)
n = dev.circuit_size()
names = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add", "w_4_value")
vals = ("w_l_value", "w_r_value", "w_o_value")
t = {k: torch.empty((n, 4), dtype=torch.int64, device="cuda:0") for k in names + vals}
t["w_4"] = torch.empty((n,), dtype=torch.int64, device="cuda:0")
lib = _lib.load()


def run(keys):
    fc = _lib.FullColumnsC(**{k: (t[k].data_ptr() if k in keys else None) for k in t})
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert lib.pg_composer_materialize(dev._h, C.byref(fc)) == 0
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    wr = sum(32 if k != "w_4" else 8 for k in keys) * n
    rd = sum(8 + 32 for k in keys if k in vals) * n
    print(json.dumps({"columns": list(keys), "ms": round(best * 1e3, 3), "written_GB": round(wr / 1e9, 2), "read_GB_algorithmic": round(rd / 1e9, 2),
                      "write_TBps": round(wr / best / 1e12, 2), "total_TBps": round((wr + rd) / best / 1e12, 2)}), flush=True)


run(names + vals + ("w_4",))
run(names + ("w_4",))
run(vals)
if len(sys.argv) <= 2:
    run(("w_l_value",))
    run(("w_o_value",))
    run(("q_arith",))
