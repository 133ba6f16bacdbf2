"""Times the 'next' rows of SURVEY section 8f on a large batched circuit: f1 pg_composer_materialize, f2
pg_composer_permutation, and pg_composer_check, on a composer holding `batch` x range_check(0, 2^254).
usage: python tools/time_next_rows.py [log2_batch ...]"""
import json
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth


def main():
    eng = pg.Engine(0)
    S = pg.BlsScalar.from_int
    for lg in [int(a) for a in sys.argv[1:]] or [12, 14, 16]:
        batch = 1 << lg
        rows, nvars = 3 + batch * 1031, 5 + batch * 1034
        dev = pg.StandardComposer(eng, rows + 8, nvars + 8)
        wit = torch.from_numpy(synth.uniform_below(batch, 2**254 + 2**250, seed=1).view(np.int64)).to("cuda:0")
        torch.cuda.synchronize()
        t = time.perf_counter(); dev.range_check_batch(S(0), S(2**254), wit); torch.cuda.synchronize(); t_emit = time.perf_counter() - t
        n = dev.circuit_size()
        padded = 1 << (n - 1).bit_length()
        out = {"batch": batch, "rows": n, "padded": padded, "emit_ms": round(t_emit * 1e3, 2)}
        for name, fn in (("check", dev.check), ("materialize", dev.materialize), ("permutation", lambda: dev.permutation(padded))):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
                del r
            out[name + "_ms"] = round(best * 1e3, 2)
            out[name + "_rows_per_s"] = float("%.3g" % (n / best))
        print(json.dumps(out), flush=True)
        del dev, wit
        torch.cuda.empty_cache()
    # ragged items: 2^17 x max_bound with one 253-bit bound per item (BASELINE config 4's shape)
    import bench
    batch = 1 << 17
    mr, wt = bench.c4_inputs(batch)
    dev = pg.StandardComposer(eng, 3 + 515 * batch + 8, 5 + 517 * batch + 8)
    t = time.perf_counter()
    dev.max_bound_ragged_batch(torch.from_numpy(mr.view(np.int64)).to("cuda:0"), torch.from_numpy(wt.view(np.int64)).to("cuda:0"))
    torch.cuda.synchronize()
    t_emit = time.perf_counter() - t
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter(); r = dev.permutation(padded); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        del r
    print(json.dumps({"gadget": "max_bound, per-item bounds", "batch": batch, "rows": n, "append_ms_incl_upload": round(t_emit * 1e3, 2),
                      "permutation_ms": round(best * 1e3, 2), "permutation_rows_per_s": float("%.3g" % (n / best))}), flush=True)
    del dev
    torch.cuda.empty_cache()
    # small items: 2^22 x maybe_equal on Variables allocated before (3 rows / 3 Variables per item, linked several
    # hundred items per workgroup; both inputs of every item go through the sorted list)
    batch = 1 << 22
    dev = pg.StandardComposer(eng, 3 + 3 * batch + 8, 5 + 5 * batch + 8)
    a = torch.from_numpy(synth.random_scalars(batch, 5).view(np.int64)).to("cuda:0")
    fa, fb = dev.add_input_batch(a), dev.add_input_batch(a)
    av = torch.arange(fa, fa + batch, dtype=torch.int64, device="cuda:0")
    dev.maybe_equal_batch(av, av + batch)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter(); r = dev.permutation(padded); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        del r
    print(json.dumps({"gadget": "maybe_equal", "batch": batch, "rows": n, "permutation_ms": round(best * 1e3, 2),
                      "permutation_rows_per_s": float("%.3g" % (n / best))}), flush=True)


if __name__ == "__main__":
    main()
