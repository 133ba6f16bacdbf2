"""tools/mat_variants.py LIB... -- on the GPU box: pg_composer_materialize on a 270 M-row composer with each variant library
(tools/variants/lib_*.so), one child process per library; best and median of 7 calls, ms."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, lg):
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import torch
    from plonk_gadgets_amd import _lib
    _lib.LIB_PATH = lib
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import synth
    eng = pg.Engine(0)
    S = pg.BlsScalar.from_int
    batch = 1 << lg
    dev = pg.StandardComposer(eng, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    wit = torch.from_numpy(synth.random_scalars(batch, seed=1).view(np.int64)).to("cuda:0")
    dev.range_check_batch(S(0), S(2**254), wit)
    n = dev.circuit_size()
    padded = 1 << (n - 1).bit_length()
    res = {}
    for name, fn in (("materialize", dev.materialize), ("permutation", lambda: dev.permutation(padded))):
        ms = []
        for _ in range(7):
            torch.cuda.synchronize()
            t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
            del r
        ms.sort()
        res[name] = {"best": round(ms[0], 2), "median": round(ms[3], 2)}
    print(json.dumps({"lib": os.path.basename(lib), "rows": n, **res}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        for lib in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, "--child", os.path.abspath(lib), "18"], check=False)
