import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import plonk_gadgets_amd as pg
S = pg.BlsScalar.from_int
eng = pg.Engine(0)
dev = pg.StandardComposer(eng, 1 << 21, 1 << 21)
mn, mx = S(0), S(2**254)
for _ in range(20):
    pg.range_check(dev, mn, mx, pg.AllocatedScalar.allocate(dev, S(12345)))
torch.cuda.synchronize()
N = 1000
t = time.perf_counter()
for i in range(N):
    pg.range_check(dev, mn, mx, pg.AllocatedScalar.allocate(dev, S(i)))
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("single range_check calls: %.1f us per call, %.3g constraints/s" % (dt / N * 1e6, N * 1031 / dt))
t = time.perf_counter()
for i in range(N):
    dev.boolean_gate(5)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("single gate calls: %.1f us per call" % (dt / N * 1e6))
