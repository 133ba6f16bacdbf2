"""GPU: the f-rows at the sizes bench.py times them, EVERY word against the CPU oracle -- no sampling.

pg_composer_permutation (sigma, four columns over the padded domain) and the eleven arrays of pg_composer_materialize are
computed on the device for whole circuits -- bench.py's own `next_rows` composer (2^18 x (allocate + range_check(0, 2^254)):
270 270 467 rows, sigma padded to 2^29), and the other shapes that have closed forms of their own (witnesses allocated before the
call, per-item bounds, the fused scalar mix, the reference tests' loop through the command queue) -- and compared word for word
with what oracle/fast.c computes for the same circuit on the host:

    tests/frows_oracle.HostCircuit   the circuit's four wire columns and its assignment table, assembled from the threaded gadget
                                     forms (pinned to oracle/gadgets.c + oracle/composer.c by tests/test_oracle_fast.py)
    oracle_sigma_fast_*              sigma from the wire columns (== composer_sigma, Permutation::compute_sigma_permutations restated)
    oracle_materialize_fast          the constant selector columns, w_4 and the wire-VALUE columns (== the composer's own)

streamed chunk by chunk through pinned host memory while the oracle's threads write the next chunk.  The composer's own wire
columns and table are compared with the host circuit first (so the f-rows are checked on the circuit they were computed from).
A difference is reported as array / row / limb with both values; every test flips one bit in each device array and requires
exactly that report."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

import plonk_gadgets_amd as pg
from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"
S = pg.BlsScalar.from_int
MAT_SCALARS = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add", "w_l_value", "w_r_value",
               "w_o_value", "w_4_value")
SIGMA = ("sigma[w_l]", "sigma[w_r]", "sigma[w_o]", "sigma[w_4]")


def oracle_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(DEV)


def require_host_memory(rows, variables):
    """skip (and say so) rather than let the host's out-of-memory killer take the box"""
    from tests.frows_oracle import host_bytes, host_memory_available
    need, have = host_bytes(rows, variables), host_memory_available()
    if have < need:
        pytest.skip("the oracle's side of a %d-row circuit needs %.0f GB of host memory, %.0f are available" % (rows, need / 1e9, have / 1e9))


def release_hbm():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def engine():
    e = pg.Engine(0)
    yield e
    e.close()


class Stager:
    """two sets of pinned host buffers (the oracle writes one while the other is uploaded and compared) + one device stage,
    per array name, grown on demand and kept for the module's tests (pinning is the expensive part)"""

    def __init__(self):
        self.pinned = [{}, {}]
        self.stage = {}

    def views(self, k: int, spec: dict):
        out = {}
        for name, words in spec.items():
            t = self.pinned[k].get(name)
            if t is None or t.numel() < words:
                t = self.pinned[k][name] = torch.empty((words,), dtype=torch.int64, pin_memory=True)
            out[name] = t
        return out

    def device(self, name: str, words: int):
        t = self.stage.get(name)
        if t is None or t.numel() < words:
            t = self.stage[name] = torch.empty((words,), dtype=torch.int64, device=DEV)
        return t


@pytest.fixture(scope="module")
def stager():
    s = Stager()
    yield s
    s.pinned = s.stage = None
    release_hbm()


def describe(name, got, exp, r0):
    d = (got.reshape(-1) != exp.reshape(-1)).nonzero()
    k = int(d[0])
    per = got.shape[1] if got.dim() == 2 else 1
    g, e = int(got.reshape(-1)[k]) & (2**64 - 1), int(exp.reshape(-1)[k]) & (2**64 - 1)
    return f"{name}: row {r0 + k // per} limb {k % per}: device {g:#018x}, oracle {e:#018x} ({int(d.numel())} differing words in this chunk)"


def stream_check(stager, arrays: dict, total: int, chunk: int, produce, descending=False, only=None):
    """arrays: name -> device tensor [total] or [total, 4]; produce(r0, r1, views) fills the flat uint64 numpy views (one per
    name, at least (r1 - r0) * words-per-row long) with rows [r0, r1) of the oracle's arrays.  Chunks go up (or down: sigma) the
    rows; `only` = the index of the one chunk to do.  Returns the words compared; AssertionError names array / row / limb."""
    per = {k: (t.shape[1] if t.dim() == 2 else 1) for k, t in arrays.items()}
    ranges = [(r0, min(total, r0 + chunk)) for r0 in range(0, total, chunk)]
    if descending:
        ranges.reverse()
    if only is not None:
        ranges = [ranges[only]]
    spec = {k: chunk * per[k] for k in arrays}
    sets = [stager.views(0, spec), stager.views(1, spec)]
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    ready = [threading.Semaphore(0), threading.Semaphore(0)]
    failure = []

    def producer():
        try:
            for k, (r0, r1) in enumerate(ranges):
                free[k % 2].acquire()
                if failure:
                    return
                produce(r0, r1, {name: t.numpy().view(np.uint64) for name, t in sets[k % 2].items()})
                ready[k % 2].release()
        except BaseException as e:  # (the consumer must not wait for a chunk that will never come)
            failure.append(e)
            for r in ready:
                r.release()

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    words = 0
    try:
        for k, (r0, r1) in enumerate(ranges):
            ready[k % 2].acquire()
            if failure:
                raise failure[0]
            m = r1 - r0
            bad = torch.zeros((), dtype=torch.bool, device=DEV)
            views = {}
            for name, t in arrays.items():
                w = m * per[name]
                st = stager.device(name, spec[name])
                st[:w].copy_(sets[k % 2][name][:w], non_blocking=True)
                exp = st[:w].view(m, per[name]) if per[name] > 1 else st[:w]
                got = t[r0:r1]
                bad |= (got != exp).any()
                views[name] = (got, exp)
                words += w
            if bool(bad):
                for name, (got, exp) in views.items():
                    if not torch.equal(got, exp):
                        failure.append(AssertionError(f"rows [{r0}, {r1}): " + describe(name, got, exp, r0)))
                        raise failure[0]
            free[k % 2].release()
    finally:
        failure.append(None)
        for f in free:
            f.release()
        th.join(timeout=120)
    return words


def flip_and_find(tensor, name, row, limb, run):
    """ONE flipped bit in the device's output is reported at its array / row / limb"""
    cell = tensor[row, limb] if tensor.dim() == 2 else tensor[row]
    old = int(cell)
    cell.fill_(old ^ (1 << 13))
    try:
        with pytest.raises(AssertionError) as e:
            run()
        assert f"{name}: row {row} limb {limb}:" in str(e.value) and "(1 differing words" in str(e.value), str(e.value)
    finally:
        cell.fill_(old)


def check_with_flips(stager, arrays, total, chunk, produce, flips, descending=False):
    """stream_check over everything, then the one-bit self-test: for every (name, row, limb) of `flips` one bit of the DEVICE array is
    flipped and the chunk that holds it compared again -- against the oracle's chunk as the full pass produced it (kept on the host)"""
    per = {k: (t.shape[1] if t.dim() == 2 else 1) for k, t in arrays.items()}
    wanted, saved = {row // chunk for _, row, _ in flips}, {}

    def produce_and_keep(r0, r1, v):
        produce(r0, r1, v)
        if r0 // chunk in wanted:
            saved[r0 // chunk] = {name: v[name][:(r1 - r0) * per[name]].copy() for name in arrays}
    words = stream_check(stager, arrays, total, chunk, produce_and_keep, descending=descending)
    n_chunks = -(-total // chunk)
    for name, row, limb in flips:
        ci = row // chunk

        def replay(r0, r1, v):
            assert r0 == ci * chunk
            for k in arrays:
                v[k][:(r1 - r0) * per[k]] = saved[ci][k]
        only = n_chunks - 1 - ci if descending else ci
        flip_and_find(arrays[name], name, row, limb, lambda: stream_check(stager, arrays, total, chunk, replay, descending=descending, only=only))
    return words


def check_f_rows(stager, comp, host, sigma_chunk=1 << 22, mat_chunk=1 << 20, flip_rows=(), materialize=True):
    """every word of the composer's wire columns and table, of pg_composer_permutation over the padded domain and of the eleven
    pg_composer_materialize arrays against the host circuit's; flip_rows: rows at which the one-bit self-tests are made"""
    import ctypes as C
    from plonk_gadgets_amd import _lib
    n, nv = comp.circuit_size(), comp.num_variables()
    assert (n, nv) == (host.n, host.nv), ((n, nv), (host.n, host.nv))
    padded = 1 << (n - 1).bit_length()
    flip_rows = list(flip_rows) or [n // 2, n // 3, n - 1, 0]
    assert all(0 <= r < n for r in flip_rows)
    words = {}
    # -- the circuit itself: wire columns and assignments (the f-rows below are functions of exactly these)
    cols = comp.device_columns()
    hw, table = host.wires(), host.table()

    def wires_chunk(r0, r1, v):
        for k, name in enumerate(("w_l", "w_r", "w_o")):
            v[name][:r1 - r0] = hw[k][r0:r1]
    words["wires"] = stream_check(stager, {k: getattr(cols, k) for k in ("w_l", "w_r", "w_o")}, n, 1 << 23, wires_chunk)

    def table_chunk(r0, r1, v):
        v["var_values"][:4 * (r1 - r0)] = table[r0:r1].reshape(-1)
    words["var_values"] = stream_check(stager, {"var_values": cols.var_values}, nv, 1 << 22, table_chunk)
    del cols
    # -- f2: sigma
    sigma = torch.full((4, padded), -1, dtype=torch.int64, device=DEV)
    guard = torch.full((64,), -1, dtype=torch.int64, device=DEV)
    st = comp._lib.pg_composer_permutation(comp._h, padded, sigma.data_ptr())
    assert st == 0, st
    torch.cuda.synchronize()
    sf = host.sigma_plan(padded)

    def sigma_chunk_of(r0, r1, v):
        sf.chunk(r0, r1, out=[v[name] for name in SIGMA])
    flips = [(name, flip_rows[w % len(flip_rows)], 0) for w, name in enumerate(SIGMA)]
    words["sigma"] = check_with_flips(stager, {name: sigma[w] for w, name in enumerate(SIGMA)}, padded, sigma_chunk, sigma_chunk_of,
                                      flips, descending=True)
    sf.close()
    assert words["sigma"] == 4 * padded and bool((guard == -1).all())
    del sigma
    release_hbm()
    if not materialize:
        return words
    # -- f1: the materialised columns
    m = {k: torch.full((n, 4), -1, dtype=torch.int64, device=DEV) for k in MAT_SCALARS}
    m["w_4"] = torch.full((n,), -1, dtype=torch.int64, device=DEV)
    fc = _lib.FullColumnsC(**{k: v.data_ptr() for k, v in m.items()})
    st = comp._lib.pg_composer_materialize(comp._h, C.byref(fc))
    assert st == 0, st
    torch.cuda.synchronize()
    flips = [(name, flip_rows[i % len(flip_rows)], (i % 4) if name != "w_4" else 0) for i, name in enumerate(list(MAT_SCALARS) + ["w_4"])]
    words["materialize"] = check_with_flips(stager, m, n, mat_chunk, lambda r0, r1, v: host.materialize(r0, r1, out=v), flips)
    assert words["materialize"] == 41 * n
    del m
    release_hbm()
    return words


def test_bench_next_rows_composer_every_word(engine, stager):
    """the composer bench.py's `next_rows` times (bench.next_rows_secondary): 2^18 x (allocate + range_check(0, 2^254)) appended in
    one call -- 270 270 467 rows, 271 056 901 Variables; sigma over the padded 2^29 rows (17.2 GB: position indices up to 2^31)
    and all eleven materialised arrays (88.6 GB), every word.  The loop: /root/reference/src/allocated_scalar.rs:27,
    src/range.rs:27-158."""
    from tests.frows_oracle import HostCircuit
    release_hbm()
    batch = 1 << 18
    free, _ = torch.cuda.mem_get_info()
    if free < (200 << 30):
        pytest.skip("not enough free HBM for the 270 M-row composer and its f-rows")
    require_host_memory(3 + batch * 1031, 5 + batch * 1034)
    comp = pg.StandardComposer(engine, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    wit = synth.random_scalars(batch, seed=synth.SEED + 2)          # (bench.py's witnesses for this composer)
    res = comp.range_check_batch(S(0), S(2**254), dev(wit))
    host = HostCircuit(3 + batch * 1031, 5 + batch * 1034, threads=oracle_threads())
    ores = host.range_check_batch(0, 2**254, wit)
    assert torch.equal(res, dev(ores))
    n = comp.circuit_size()
    assert n == 270_270_467 and (1 << (n - 1).bit_length()) == 1 << 29
    words = check_f_rows(stager, comp, host, flip_rows=(2, 3 + 1031 * 70_001 + 517, n - 1, 3 + 1031 * (batch - 1)))
    assert words["sigma"] * 8 == 17_179_869_184 and words["materialize"] * 8 == 328 * n
    comp.close()
    del comp, host
    release_hbm()


def test_allocated_witnesses_every_word(engine, stager):
    """the reference's own signature (/root/reference/src/range.rs:27-32: the witness is an AllocatedScalar the caller made):
    2^17 witnesses allocated first (pg_composer_add_input_batch), then range_check(0, 2^254) over them
    (pg_composer_range_check_allocated_batch): 135 M rows; every item's witness Variable lies outside the item, so its positions
    go through the permutation's sparse list and its value through the materialisation's loader wave"""
    from tests.frows_oracle import HostCircuit
    release_hbm()
    batch = 1 << 17
    G, V = 1031, 1033
    comp = pg.StandardComposer(engine, 3 + batch * G + 8, 5 + batch * (V + 1) + 8)
    wit = synth.random_scalars(batch, seed=0xA110C)
    d_wit = dev(wit)
    first = comp.add_input_batch(d_wit)
    wv = torch.arange(first, first + batch, dtype=torch.int64, device=DEV)
    res = comp.range_check_allocated_batch(S(0), S(2**254), wv, d_wit)
    host = HostCircuit(3 + batch * G, 5 + batch * (V + 1), threads=oracle_threads())
    assert host.add_input_batch(wit) == first
    ores = host.range_check_allocated_batch(0, 2**254, np.arange(first, first + batch, dtype=np.uint64), wit)
    assert torch.equal(res, dev(ores))
    n = comp.circuit_size()
    check_f_rows(stager, comp, host, flip_rows=(3, 3 + G * 4097, 3 + G * 100_000 + 1030, n - 1))
    comp.close()
    del comp, host
    release_hbm()


def test_per_item_bounds_every_word(engine, stager):
    """a C4-shaped composer: 2^17 x (allocate + max_bound(a 253-bit bound of its own)) (pg_composer_max_bound_ragged_batch;
    /root/reference/src/range.rs:82-113) -- items of different lengths, their places from the call's prefix sums"""
    import bench
    from tests.frows_oracle import HostCircuit
    release_hbm()
    batch = 1 << 17
    mr, wt = bench.c4_inputs(batch, seed=0xC4 + 6)
    comp = pg.StandardComposer(engine, 3 + batch * 515 + 8, 5 + batch * 517 + 8)
    res, nb = comp.max_bound_ragged_batch(dev(mr), dev(wt))
    host = HostCircuit(3 + batch * 515, 5 + batch * 517, threads=oracle_threads())
    ores, onb = host.max_bound_ragged_batch(mr, wt)
    assert torch.equal(res, dev(ores)) and torch.equal(nb.to(torch.int64), dev(onb))
    n = comp.circuit_size()
    check_f_rows(stager, comp, host, flip_rows=(5, n // 2 + 1, n - 3, n // 7))
    comp.close()
    del comp, host
    release_hbm()


@pytest.mark.parametrize("form", ["complete", "failing_items"])
def test_scalar_mix_composer_every_word(engine, stager, form):
    """a C3-shaped composer: 2^20 fused items (five add_input + is_non_zero + conditionally_select_one + maybe_equal:
    /root/reference/src/scalar.rs:36-140) appended by pg_composer_scalar_mix_batch; `failing_items`: with v = 0 sprinkled in
    (is_non_zero stops after its first row, src/scalar.rs:69-79: the segment is ragged and zero_var sits on an output wire)"""
    import bench
    from tests.frows_oracle import HostCircuit
    release_hbm()
    batch = 1 << 20
    v, y, s, a, b = bench.mix_inputs(batch, seed=0xC3 + 6)
    zeros = []
    if form == "failing_items":
        zeros = sorted(set(range(11, batch, 5003)) | set(range(300_000, 300_050)) | {0, 255, 256, batch - 1})
        v[zeros] = 0
    comp = pg.StandardComposer(engine, 3 + batch * 10 + 8, 5 + batch * 15 + 8)
    res, err, nerr = comp.scalar_mix_batch(*[dev(x) for x in (v, y, s, a, b)])
    assert nerr == len(zeros)
    host = HostCircuit(3 + batch * 10, 5 + batch * 15, threads=oracle_threads())
    ores, oerr = host.scalar_mix_batch(v, y, s, a, b)
    assert torch.equal(res, dev(ores)) and torch.equal(err, torch.from_numpy(oerr).to(DEV))
    n = comp.circuit_size()
    assert n == 3 + 10 * batch - 2 * len(zeros)
    check_f_rows(stager, comp, host, flip_rows=(4, n // 2, n - 1, 3 + 10 * 12345))
    comp.close()
    del comp, host
    release_hbm()


def test_reference_loop_through_the_queue_every_word(engine, stager):
    """the reference tests' loop, call by call through the command queue (/root/reference/tests/range_gadgets_tests.rs:29-44):
    4096 x { AllocatedScalar::allocate; range_check; constrain_to_constant(res, outcome, None) } -- a flush sends the gadget calls
    out as one strided launch and the gates as one run.  Two runs: range_check(0, 2^254) as bench.py's `single_calls` has it
    (n = 255, where every witness passes: SURVEY 8d) and range_check(0, 2^253) (n = 254: both outcomes); the outcome each result
    is constrained to is the oracle's assignment of that result, so the circuit is satisfied only if the device agrees; 4.2 M rows"""
    from tests.frows_oracle import HostCircuit
    release_hbm()
    calls = 4096
    ints = [(int(x) << 192) % (2**253 + 2**252) for x in synth.splitmix64(calls, 0x100B)]   # a third of them above 2^253
    wit = synth.scalars_from_ints(ints)
    host = HostCircuit(3 + calls * 1032, 5 + calls * 1034, threads=oracle_threads())
    runs = [(2**254, 0, calls // 2), (2**253, calls // 2, calls)]
    ores = np.concatenate([host.range_check_loop_with_constrain(0, mx, wit[lo:hi]) for mx, lo, hi in runs])
    one = np.array(synth.mont(1), dtype=np.uint64)
    outcome = (host.table()[ores.astype(np.int64)] == one).all(axis=1)
    assert bool(outcome[:calls // 2].all()) and 0 < int(outcome[calls // 2:].sum()) < calls // 2
    comp = pg.StandardComposer(engine, 3 + calls * 1032 + 8, 5 + calls * 1034 + 8)
    comp.queue(True)
    res = []
    for mx, lo, hi in runs:
        mn, mxs = S(0), S(mx)
        for i in range(lo, hi):
            r = pg.range_check(comp, mn, mxs, pg.AllocatedScalar.allocate(comp, S(ints[i])))
            comp.constrain_to_constant(r, S(int(outcome[i])), None)
            res.append(r)
    comp.sync()
    assert res == [int(r) for r in ores]
    assert comp.check() == -1
    n = comp.circuit_size()
    assert n == host.n == 3 + (calls // 2) * (1032 + 1028)
    check_f_rows(stager, comp, host, sigma_chunk=1 << 20, mat_chunk=1 << 19, flip_rows=(3 + 1031, 3 + 1032 * 2000 + 1031, n - 1, 7))
    comp.close()
    del comp, host
    release_hbm()


def test_sigma_padded_to_2_pow_30(engine, stager):
    """pg_composer_permutation over a padded domain of 2^30 rows: a composer of 2^19 + 2^12 x (allocate + range_check(0, 2^254)) =
    544 763 779 rows; sigma's 4 x 2^30 entries (34.4 GB) hold positions up to 2^32 and lie at byte offsets up to 2^35 -- every
    word against the oracle's (the materialised columns of a composer this size do not fit beside it: sigma and the circuit only)"""
    from tests.frows_oracle import HostCircuit
    release_hbm()
    batch = (1 << 19) + (1 << 12)
    free, _ = torch.cuda.mem_get_info()
    if free < (190 << 30):
        pytest.skip("not enough free HBM for a 545 M-row composer and a 34-GB sigma")
    require_host_memory(3 + batch * 1031, 5 + batch * 1034)
    comp = pg.StandardComposer(engine, 3 + batch * 1031 + 8, 5 + batch * 1034 + 8)
    wit = synth.random_scalars(batch, seed=synth.SEED + 30)
    res = comp.range_check_batch(S(0), S(2**254), dev(wit))
    host = HostCircuit(3 + batch * 1031, 5 + batch * 1034, threads=oracle_threads())
    ores = host.range_check_batch(0, 2**254, wit)
    assert torch.equal(res, dev(ores))
    n = comp.circuit_size()
    assert (1 << (n - 1).bit_length()) == 1 << 30
    words = check_f_rows(stager, comp, host, flip_rows=(n - 1, (1 << 29) + 17, 3 + 1031 * 400_000 + 600, 1), materialize=False)
    assert words["sigma"] == 1 << 32
    comp.close()
    del comp, host
    release_hbm()


def test_small_gadgets_and_gate_batches_every_word(engine, stager):
    """the batched appends on EXISTING Variables -- conditionally_select_zero / _one, maybe_equal, is_non_zero (with items that stop at
    their error: ragged; and without: uniform) (/root/reference/src/scalar.rs:21-140) and the composer's add / mul / poly_gate /
    constrain_to_constant / boolean_gate over arrays -- 2^18 items each, every batch's inputs drawn from ALL Variables that exist by
    then (the allocated inputs, zero_var, results of the earlier batches).  Their sigma comes from the kinds' wire tables
    (csrc/permutation.hpp, perm_template_kernel: own Variables' cycles in closed form, Variables from elsewhere through the sorted
    list at closed-form slots); the wires, the assignments, sigma and the materialised columns, every word."""
    from tests.frows_oracle import HostCircuit
    release_hbm()
    batch = 1 << 18
    rng = np.random.default_rng(0x5A11)
    Q = synth.Q
    ints = [0, 0, 1, 1, 7, 7] + [int(x) for x in synth.splitmix64(batch - 6, 0xB00)]
    scal = synth.scalars_from_ints([x % (1 << 200) for x in ints])
    scal[100:batch:97] = 0                                   # zeros sprinkled in: is_non_zero stops there
    comp = pg.StandardComposer(engine, 3 + 20 * batch, 5 + 18 * batch)
    host = HostCircuit(3 + 20 * batch, 5 + 18 * batch, threads=oracle_threads())
    first = comp.add_input_batch(dev(scal))
    assert host.add_input_batch(scal) == first
    tv = lambda a: torch.from_numpy(a.astype(np.int64)).to(DEV)
    pick = lambda: rng.integers(0, host.nv, size=batch, dtype=np.uint64)

    def same(dev_res, host_res):
        assert torch.equal(dev_res, dev(host_res))
    a = pick()
    a[:4] = [first, 0, first + 1, first + 100]               # value 0 four times, once zero_var itself
    err, nerr = comp.is_non_zero_batch(tv(a))
    _, oerr = host.small_batch("is_non_zero", a)
    assert nerr == int(oerr.sum()) > batch // 200 and torch.equal(err, torch.from_numpy(oerr).to(DEV))
    nz = np.flatnonzero(host.table()[:first + batch].any(axis=1)).astype(np.uint64)   # Variables whose value is not 0: no item fails
    a = nz[rng.integers(0, len(nz), size=batch)]
    err, nerr = comp.is_non_zero_batch(tv(a))
    _, oerr = host.small_batch("is_non_zero", a)
    assert nerr == 0 and not oerr.any()
    a, b = pick(), pick()
    a[:2], b[:2] = [0, first + 2], [first + 3, 0]
    same(comp.conditionally_select_zero_batch(tv(a), tv(b)), host.small_batch("select_zero", a, b))
    a, b = pick(), pick()
    same(comp.conditionally_select_one_batch(tv(a), tv(b)), host.small_batch("select_one", a, b))
    a, b = pick(), pick()
    a[:3], b[:3] = [first + 4, first, 0], [first + 5, first + 1, 0]        # equal assignments on different Variables; zero_var twice
    b[1000:2000] = a[1000:2000]                                               # the same Variable on both inputs
    same(comp.maybe_equal_batch(tv(a), tv(b)), host.small_batch("maybe_equal", a, b))
    q = [int(x) % Q for x in synth.splitmix64(5, 0xC0FFEE)]
    a, b = pick(), pick()
    same(comp.add_batch(S(q[1]), tv(a), S(q[2]), tv(b), S(q[4])), host.small_batch("add", a, b, selectors=(0, q[1], q[2], Q - 1, q[4])))
    a, b = pick(), pick()
    same(comp.mul_batch(S(q[0]), tv(a), tv(b), S(q[4])), host.small_batch("mul", a, b, selectors=(q[0], 0, 0, Q - 1, q[4])))
    a, b, c3 = pick(), pick(), pick()
    comp.poly_gate_batch(tv(a), tv(b), tv(c3), *[S(x) for x in q])
    host.small_batch("rows", a, b, c3, selectors=q)
    a = pick()
    comp.constrain_to_constant_batch(tv(a), S(7))
    host.small_batch("rows", a, a, a, selectors=(0, 1, 0, 0, Q - 7))
    a = pick()
    comp.boolean_gate_batch(tv(a))
    host.small_batch("rows", a, a, a, selectors=(1, 0, 0, Q - 1, 0))
    n = comp.circuit_size()
    assert n == host.n > 17 * batch
    check_f_rows(stager, comp, host, sigma_chunk=1 << 21, mat_chunk=1 << 19, flip_rows=(3, n // 2, n - 1, 3 + 3 * batch // 2))
    comp.close()
    del comp, host
    release_hbm()
