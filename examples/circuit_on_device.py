#!/usr/bin/env python3
"""A circuit built, checked and made prover-ready without leaving the GPU.

    python examples/circuit_on_device.py [log2_witnesses]

The reference's loop
    for w in witnesses { let a = AllocatedScalar::allocate(composer, w); results.push(range_check(composer, min, max, a)); }
is one batched append on the device-resident composer; single calls with the reference's signatures mix in freely.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    batch = 1 << lg
    S = pg.BlsScalar.from_int
    engine = pg.Engine(0)
    composer = pg.StandardComposer(engine)
    composer.auto_grow()                                   # grow like the reference's Vecs ...
    lay = engine.range_check_layout(S(0), S(2**254), batch)
    composer.reserve(lay.n_gates + 64, lay.n_vars + 64)    # ... but the big append's size is known: make room once

    # witnesses arrive as canonical little-endian bytes (BlsScalar::to_bytes): convert in bulk on the device
    canonical = torch.from_numpy(synth.splitmix64(4 * batch, 1).reshape(batch, 4).copy().view(np.int64)).to("cuda:0")
    canonical[:, 3] &= (1 << 60) - 1                       # keep them below q
    witnesses, _, bad = engine.scalars_from_canonical(canonical)
    assert bad == 0

    t = time.perf_counter()
    results = composer.range_check_batch(S(0), S(2**254), witnesses)          # allocate + range_check, per witness
    first = pg.AllocatedScalar.allocate(composer, S(7))                       # single calls, reference signatures
    inside = pg.range_check(composer, S(5), S(10), first)
    both = pg.conditionally_select_zero(composer, inside, int(results[0]))    # inside AND first batched result
    composer.constrain_to_constant(both, S(1), None)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t
    n = composer.circuit_size()
    assert composer.check() == -1                          # every row satisfies its gate equation

    t = time.perf_counter()
    padded = 1 << (n - 1).bit_length()
    full = composer.materialize()                          # constant columns, fourth wire, wire values
    sigma = composer.permutation(padded)                   # copy permutation, int64[4, padded]
    torch.cuda.synchronize()
    t_ready = time.perf_counter() - t
    print("%d witnesses -> %d rows, %d variables: built in %.2f ms (%.3g constraints/s), prover-ready in another %.2f ms"
          % (batch, n, composer.num_variables(), t_build * 1e3, n / t_build, t_ready * 1e3))
    assert full["q_arith"].shape == (n, 4) and sigma.shape == (4, padded)


if __name__ == "__main__":
    main()
