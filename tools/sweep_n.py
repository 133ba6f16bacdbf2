#!/usr/bin/env python3
"""range_check throughput over ladder lengths and in-range fractions (1 GPU): GB/s of algorithmic bytes, pre-pass included"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth

dev = torch.device("cuda", 0)
eng = pg.Engine(0)
stream = torch.cuda.current_stream(dev)
TARGET_BYTES = 48e9
for mn, mx in ((0, 2), (50_000, 250_000), (0, 2**64), (2**126, 2**127 + 1), (0, 2**254)):
    mnS, mxS = pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx)
    lay1 = eng.range_check_layout(mnS, mxS, 1)
    per_item = lay1.gates_per_item * 184 + lay1.vars_per_item * 32
    batch = 1 << int(np.floor(np.log2(TARGET_BYTES / per_item)))
    for frac_in in (1.0, 0.5, 0.0):
        k = int(batch * frac_in)
        span = max(mx - mn, 1)
        wit = synth.random_scalars(batch, 11)
        if k:
            # in-range witnesses: small canonical values mn + r (built from 64-bit draws)
            r = synth.splitmix64(k, 12) % np.uint64(min(span, 2**63))
            wit[:k] = synth.scalars_from_ints([mn + int(x) for x in r[:4096]] * (k // 4096 + 1))[:k] if k >= 4096 else \
                synth.scalars_from_ints([mn + int(x) for x in r])
        w = torch.from_numpy(np.ascontiguousarray(wit).view(np.int64)).to(dev)
        lay = eng.range_check_layout(mnS, mxS, batch)
        cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, dev)
        res = torch.empty((batch,), dtype=torch.int64, device=dev)
        eng.range_check_batch(mnS, mxS, w, 3, 5, out=cols, result_vars=res)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            eng.range_check_batch(mnS, mxS, w, 3, 5, out=cols, result_vars=res)
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[1]
        ones = int((cols.var_values.view(batch, lay.vars_per_item, 4)[:, -1, 0] != 0).sum())
        print(json.dumps({"n": lay.num_bits, "batch": batch, "accepted_fraction": ones / batch, "ms": round(t, 3),
                          "gbps": round(batch * per_item / t / 1e6), "constraints_per_s": f"{lay.n_gates / t * 1e3:.3e}"}), flush=True)
        del cols
        torch.cuda.empty_cache()
