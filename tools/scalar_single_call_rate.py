"""tools/scalar_single_call_rate.py -- on the GPU box: the scalar gadgets called ONE at a time (tests/scalar_gadgets_tests.rs), us per call"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_gadgets_amd as pg

S = pg.BlsScalar.from_int
eng = pg.Engine(0)
N = 2000
dev = pg.StandardComposer(eng, 1 << 20, 1 << 20)
dev.queue(True)
xs = [pg.AllocatedScalar.allocate(dev, S(3 + i)) for i in range(N)]
sel = pg.AllocatedScalar.allocate(dev, S(1))
dev.sync()
for name, fn in (("conditionally_select_zero", lambda a: pg.conditionally_select_zero(dev, a.var, sel.var)),
                 ("conditionally_select_one", lambda a: pg.conditionally_select_one(dev, a.var, sel.var)),
                 ("maybe_equal", lambda a: pg.maybe_equal(dev, a, sel)),
                 ("is_non_zero", lambda a: pg.is_non_zero(dev, a.var, a.scalar))):
    for rep in range(2):
        _, f0, l0 = dev.queue_stats()
        t = time.perf_counter()
        for a in xs:
            fn(a)
        dev.sync()
        dt = time.perf_counter() - t
        _, f1, l1 = dev.queue_stats()
    print("%-28s %7.2f us per call   (%d flushes, %d launches by flushes in the last pass)" % (name, dt / N * 1e6, f1 - f0, l1 - l0), flush=True)
assert dev.check() == -1
