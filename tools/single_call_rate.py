"""The reference-shaped path: ONE composer call at a time from the host (tests/range_gadgets_tests.rs:29-44), with the
composer's command queue on (calls recorded, flushed as few launches) and off (one launch per call).  1 GPU."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import plonk_gadgets_amd as pg

S = pg.BlsScalar.from_int
eng = pg.Engine(0)
mn, mx = S(0), S(2**254)
N = 2000
scalars = [S(1000 + i) for i in range(N)]  # built outside the timed loops: the loops time the composer, not Python's big integers
for queued in (False, True):
    dev = pg.StandardComposer(eng, 1 << 22, 1 << 22)
    dev.queue(queued)
    for s in scalars[:50]:
        pg.range_check(dev, mn, mx, pg.AllocatedScalar.allocate(dev, s))
    dev.sync()
    for rep in range(2):  # the second pass is the steady state: the first one also maps fresh HBM pages (890 MB of columns)
        t = time.perf_counter()
        for s in scalars:
            pg.range_check(dev, mn, mx, pg.AllocatedScalar.allocate(dev, s))
        dev.sync()
        dt = time.perf_counter() - t
        print("queue %-3s allocate + range_check: %6.2f us per pair, %.3g constraints/s%s" % (
            "on" if queued else "off", dt / N * 1e6, N * 1031 / dt, "" if rep else "  [first pass over fresh memory]"))
    t = time.perf_counter()
    for i in range(N):
        dev.boolean_gate(dev.zero_var)
    dev.sync()
    dt = time.perf_counter() - t
    print("queue %-3s gate calls:             %6.2f us per call" % ("on" if queued else "off", dt / N * 1e6))
    assert dev.check() == -1
    dev.close()
