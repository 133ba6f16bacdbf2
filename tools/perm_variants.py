"""tools/perm_variants.py LIB... -- on the GPU box: pg_composer_permutation (and pg_composer_materialize) on a 270 M-row composer with each
variant library, ALL IN ONE PROCESS, twice around (the same allocations come back from torch's cache, so every library meets the same
placement: between processes the call varies 3.0 ... 4.1 ms by placement alone).  best / median of 7 calls, ms."""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from plonk_gadgets_amd import _lib, synth
import plonk_gadgets_amd as pg


def one(lib, lg=18):
    _lib._lib, _lib.LIB_PATH = None, os.path.abspath(lib)
    eng = pg.Engine(0)
    S = pg.BlsScalar.from_int
    batch = 1 << lg
    dev = pg.StandardComposer(eng, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    wit = torch.from_numpy(synth.random_scalars(batch, seed=1).view(np.int64)).to("cuda:0")
    dev.range_check_batch(S(0), S(2**254), wit)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    res = {}
    for name, fn in (("permutation", lambda: dev.permutation(padded)), ("materialize", dev.materialize)):
        ms = []
        for _ in range(7):
            torch.cuda.synchronize()
            t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
            del r
        ms.sort()
        res[name] = {"best": round(ms[0], 3), "median": round(ms[3], 3)}
    print(json.dumps({"lib": os.path.basename(lib), "rows": n, **res}), flush=True)
    del dev, eng, wit
    gc.collect()
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for _ in range(2):
        for lib in sys.argv[1:]:
            one(lib)
