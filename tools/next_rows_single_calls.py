"""tools/next_rows_single_calls.py [N] -- on the GPU box: the f-rows of a circuit built the reference's way, ONE allocate + range_check at a
time through the command queue (N calls, 1031 rows each), against the same circuit appended as one batch.  Median of 5 calls, ms."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth

S = pg.BlsScalar.from_int
eng = pg.Engine(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mn, mx = S(0), S(2**254)


def timed(dev, label):
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    out = {}
    for name, fn in (("permutation", lambda: dev.permutation(padded)), ("materialize", dev.materialize)):
        ms = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3); del r
        ms.sort(); out[name] = round(ms[2], 3)
    print(label, "rows", n, out, flush=True)


dev = pg.StandardComposer(eng, 3 + N * 1031 + 8, 5 + N * 1035 + 8)
dev.queue(True)
scalars = [S(1000 + i) for i in range(N)]
t = time.perf_counter()
for s in scalars:
    pg.range_check(dev, mn, mx, pg.AllocatedScalar.allocate(dev, s))
dev.sync()
print("appended %d single calls in %.2f s" % (N, time.perf_counter() - t), flush=True)
timed(dev, "single calls:")
dev.close()
dev = pg.StandardComposer(eng, 3 + N * 1031 + 8, 5 + N * 1035 + 8)
wit = torch.from_numpy(np.ascontiguousarray(synth.scalars_from_ints([1000 + i for i in range(N)])).view(np.int64)).to("cuda:0")
dev.range_check_batch(mn, mx, wit)
timed(dev, "one batch:   ")
