import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import plonk_gadgets_amd as pg
S = pg.BlsScalar.from_int
eng = pg.Engine(0)
N = 4096
mn, mx = S(0), S(2**254)
scalars = [S(1000 + i) for i in range(N)]
one = S(1)
for pattern in ("pairs", "pairs+gate"):
    dev = pg.StandardComposer(eng, 3 + N * 1032 + 8, 5 + N * 1035 + 8)
    dev.queue(True)
    for rep in range(2):
        if rep: dev.clear_witness()
        t = time.perf_counter()
        for s in scalars:
            r = pg.range_check(dev, mn, mx, pg.AllocatedScalar.allocate(dev, s))
            if pattern != "pairs":
                dev.constrain_to_constant(r, one, None)
        dev.sync()
        dt = time.perf_counter() - t
        print(pattern, "rep", rep, "%.2f us per iteration" % (dt / N * 1e6), dev.queue_stats(), flush=True)
    n = dev.circuit_size(); padded = 1 << (n - 1).bit_length()
    for name, fn in (("permutation", lambda: dev.permutation(padded)), ("materialize", dev.materialize)):
        ms = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3); del r
        ms.sort(); print("  ", name, round(ms[2], 3), "ms for", n, "rows")
    assert dev.check() == -1
    dev.close()
