"""GPU: BASELINE configs C2, C3 and C4 at their FULL size, EVERY limb of all nine arrays against the CPU oracle -- no sampling.

The batch is emitted on the GPU exactly as bench.py emits it (bench.Workload: the very launch the bench times), then the
threaded form of the oracle (oracle/fast.c, pinned to the faithful restatement oracle/gadgets.c by tests/test_oracle_fast.py,
which follows /root/reference/src/range.rs:27-158 and src/scalar.rs:36-140 call for call) produces the same items chunk by
chunk, at the numbering of the whole batch, into pinned host memory; every chunk is uploaded and compared on the device with
the rows the GPU wrote -- five selector columns, three wire columns, the variable table, the result Variables, and for the
ragged configs the plan's ladder lengths and prefix sums.  A difference is reported as array / row / limb with both values.
While chunk k is uploaded and compared, the oracle's threads write chunk k + 1 (two sets of pinned buffers).

Cost: C2 is 233.6 GB through PCIe (about 5-10 s), C4 115.8 GB, C3 2.4 GB."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEL = ("q_m", "q_l", "q_r", "q_o", "q_c")
WIRES = ("w_l", "w_r", "w_o")
NINE = SEL + WIRES + ("var_values",)
LOG2_BATCH = 20
BATCH = 1 << LOG2_BATCH
DEV = "cuda:0"


def oracle_threads() -> int:
    """the box gives one GPU's share of its host: 16 cores (more threads than that ran slower in bench.py's sweep)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


@pytest.fixture(scope="module")
def engine():
    import plonk_gadgets_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


def release_hbm():
    """a failed comparison's traceback (pytest.raises in flip_and_find) keeps views of the columns alive in frame cycles"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(DEV)


class PinnedSet:
    """nine pinned host arrays the oracle writes into, and their numpy views"""

    def __init__(self, max_rows: int, max_vars: int):
        self.t = {k: torch.empty((4 * (max_vars if k == "var_values" else max_rows),), dtype=torch.int64, pin_memory=True)
                  for k in SEL + ("var_values",)}
        self.t.update({k: torch.empty((max_rows,), dtype=torch.int64, pin_memory=True) for k in WIRES})
        self.np = {k: v.numpy().view(np.uint64) for k, v in self.t.items()}


def describe_difference(name, got, exp, first_index):
    """array / row (of the whole batch) / limb of the first differing word, with both values"""
    d = (got.reshape(-1) != exp.reshape(-1)).nonzero()
    k = int(d[0])
    per = 4 if got.dim() == 2 else 1
    g, e = int(got.reshape(-1)[k]) & (2**64 - 1), int(exp.reshape(-1)[k]) & (2**64 - 1)
    unit = "variable" if name == "var_values" else "row"
    return (f"{name}: {unit} {first_index + k // per} limb {k % per}: device {g:#018x}, oracle {e:#018x} "
            f"({int(d.numel())} differing words in this chunk)")


def stream_compare(cols, offsets, produce, chunk_items, max_rows, max_vars, n_items=BATCH, only_chunk=None, names=NINE):
    """produce(lo, hi, pinned_numpy_views) fills the views with items [lo, hi) of the oracle's columns (rows relative to item
    lo's first row); offsets(i) -> (first row, first variable) of item i relative to the call.  Returns (chunks compared,
    words compared); raises AssertionError naming array / row / limb at the first difference."""
    sets = [PinnedSet(max_rows, max_vars), PinnedSet(max_rows, max_vars)]
    stage = {k: torch.empty_like(v, device=DEV) for k, v in sets[0].t.items()}
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    ready = [threading.Semaphore(0), threading.Semaphore(0)]
    chunks = [(lo, min(lo + chunk_items, n_items)) for lo in range(0, n_items, chunk_items)]
    if only_chunk is not None:
        chunks = [chunks[only_chunk]]
    failure = []

    def producer():
        try:
            for k, (lo, hi) in enumerate(chunks):
                free[k % 2].acquire()
                if failure:
                    return
                produce(lo, hi, sets[k % 2].np)
                ready[k % 2].release()
        except BaseException as e:  # (the consumer must not wait for a chunk that will never come)
            failure.append(e)
            for r in ready:
                r.release()

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    words = 0
    try:
        for k, (lo, hi) in enumerate(chunks):
            ready[k % 2].acquire()
            if failure:
                raise failure[0]
            (r0, v0), (r1, v1) = offsets(lo), offsets(hi)
            assert r1 - r0 <= max_rows and v1 - v0 <= max_vars, (lo, hi, r1 - r0, v1 - v0)
            bad = torch.zeros((), dtype=torch.bool, device=DEV)
            views = {}
            for name in names:
                per = 1 if name in WIRES else 4
                n, first = (v1 - v0, v0) if name == "var_values" else (r1 - r0, r0)
                stage[name][:n * per].copy_(sets[k % 2].t[name][:n * per], non_blocking=True)
                exp = stage[name][:n * per].view(n, 4) if per == 4 else stage[name][:n]
                got = getattr(cols, name)[first:first + n]
                bad |= (got != exp).any()
                views[name] = (got, exp, first)
                words += n * per
            if bool(bad):  # (one host synchronisation per chunk: the uploads from this set are complete as well)
                for name in names:
                    got, exp, first = views[name]
                    if not torch.equal(got, exp):
                        failure.append(AssertionError(f"items [{lo}, {hi}): " + describe_difference(name, got, exp, first)))
                        raise failure[0]
            free[k % 2].release()
    finally:
        failure.append(None)  # lets a producer that still waits for a buffer leave
        for f in free:
            f.release()
        th.join(timeout=60)
    return len(chunks), words


def flip_and_find(cols, name, index, limb, run_chunk):
    """the comparison itself is shown to work: ONE flipped bit in the device's output is reported at its array / row / limb"""
    col = getattr(cols, name)
    cell = col[index, limb] if col.dim() == 2 else col[index]
    old = int(cell)
    cell.fill_(old ^ (1 << 17))
    try:
        with pytest.raises(AssertionError) as e:
            run_chunk()
        unit = "variable" if name == "var_values" else "row"
        assert f"{name}: {unit} {index} limb {limb}:" in str(e.value) and "(1 differing words" in str(e.value), str(e.value)
    finally:
        cell.fill_(old)


def test_config_c2_every_limb(engine):
    """BASELINE config 2: 2^20 witnesses x (allocate + range_check(0, 2^254)), n = 255 -- all 1 081 081 856 rows and
    1 084 227 584 variables, 233.6 GB, limb for limb (/root/reference/src/range.rs:119-158: every ladder row)"""
    import bench
    from oracle import pyoracle as po
    G, V = 1031, 1034
    release_hbm()
    free, _ = torch.cuda.mem_get_info()
    if free < BATCH * (G * 184 + V * 32) + (24 << 30):
        pytest.skip("not enough free HBM for the full-size batch")
    wl = bench.Workload("c2", engine, DEV, 0, 1, LOG2_BATCH, -1)
    assert wl.n_chunks == 1 and wl.chunk == BATCH, "C2 is one launch on this card"
    for name in NINE:
        getattr(wl.cols, name).fill_(-1)  # a slot nobody writes differs from the oracle's
    wl.res.fill_(-1)
    wl.launch(0)
    torch.cuda.synchronize()
    wit = synth.random_scalars(BATCH, seed=synth.SEED)
    assert torch.equal(wl.wit, dev(wit))
    mn, mx = synth.mont(0), synth.mont(2**254)
    threads = oracle_threads()
    chunk = 1 << 12
    results = []

    def produce(lo, hi, out):
        r = po.range_check_fast(mn, mx, np.ascontiguousarray(wit[lo:hi]), threads=threads, var_base=5 + lo * V, out=out)
        assert r["num_bits"] == 255
        results.append(r["result_vars"])

    def run(only=None):
        return stream_compare(wl.cols, lambda i: (i * G, i * V), produce, chunk, chunk * G, chunk * V, only_chunk=only)

    n_chunks, words = run()
    assert n_chunks == BATCH // chunk and words * 8 == BATCH * (G * 184 + V * 32) == 233_614_344_192
    assert torch.equal(wl.res, dev(np.concatenate(results)))
    # one flipped bit anywhere is found and named
    item = 177 * chunk + 3
    results.clear()
    flip_and_find(wl.cols, "q_l", item * G + 517, 2, lambda: run(only=177))
    flip_and_find(wl.cols, "w_o", item * G + 1030, 0, lambda: run(only=177))
    flip_and_find(wl.cols, "var_values", item * V + 600, 3, lambda: run(only=177))
    wl.release()
    del wl, run, produce
    release_hbm()


def test_config_c4_every_limb(engine):
    """BASELINE config 4: 2^20 x (allocate + max_bound(random 253-bit bound)), ragged -- every row and variable (115.8 GB),
    the ladder lengths and both prefix sums (/root/reference/src/range.rs:82-113, :185-189)"""
    import bench
    from oracle import pyoracle as po
    release_hbm()
    free, _ = torch.cuda.mem_get_info()
    if free < BATCH * (515 * 184 + 517 * 32) + (16 << 30):
        pytest.skip("not enough free HBM for the full-size batch")
    wl = bench.Workload("c4", engine, DEV, 0, 1, LOG2_BATCH, -1)
    for name in NINE:
        getattr(wl.cols, name).fill_(-1)
    wl.res.fill_(-1)
    d_nb, d_roff, d_voff = wl.plan_buffers
    d_nb.fill_(0)
    d_roff.fill_(-1)
    d_voff.fill_(-1)
    wl.launch(0)  # (plans again, asynchronously, straight into the emit call)
    torch.cuda.synchronize()
    lay, nerr = engine.plan_result()
    mr, wt = bench.c4_inputs(BATCH, seed=0xC4)
    threads = oracle_threads()
    plan = po.max_bound_plan(mr, threads=threads)
    nb, roff, voff = plan
    assert (lay.n_gates, lay.n_vars, nerr) == (int(roff[-1]), int(voff[-1]), 0)
    assert wl.cols.q_m.shape[0] == lay.n_gates and wl.cols.var_values.shape[0] == lay.n_vars
    # the plan the launch made: ladder lengths and prefix sums, item by item
    assert torch.equal(d_nb.to(torch.int64), dev(nb)) and torch.equal(d_roff, dev(roff)) and torch.equal(d_voff, dev(voff))
    chunk = 1 << 13
    max_rows, max_vars = chunk * (2 * 255 + 5), chunk * (255 + 262)
    results = []

    def produce(lo, hi, out):
        results.append(po.max_bound_fast(mr, wt, plan, lo, hi, var_base=5, threads=threads, out=out)["result_vars"])

    def run(only=None):
        return stream_compare(wl.cols, lambda i: (int(roff[i]), int(voff[i])), produce, chunk, max_rows, max_vars, only_chunk=only)

    n_chunks, words = run()
    assert n_chunks == BATCH // chunk and words * 8 == lay.n_gates * 184 + lay.n_vars * 32
    assert torch.equal(wl.res, dev(np.concatenate(results)))
    results.clear()
    item = 100 * chunk + 4099  # (an item in the middle of a chunk: its first row's q_c is the one data-dependent selector)
    flip_and_find(wl.cols, "q_c", int(roff[item]), 1, lambda: run(only=100))
    flip_and_find(wl.cols, "w_r", int(roff[item + 1]) - 1, 0, lambda: run(only=100))
    flip_and_find(wl.cols, "var_values", int(voff[item]) + 300, 0, lambda: run(only=100))
    wl.release()
    del wl, run, produce
    release_hbm()


@pytest.mark.parametrize("form", ["bench", "failing_items"])
def test_config_c3_every_limb(engine, form):
    """BASELINE config 3: 2^20 fused items (5 add_input + is_non_zero + conditionally_select_one + maybe_equal), every row
    and variable.  `bench`: bench.Workload's launch on bench.py's inputs (no item fails).  `failing_items`: the same call on
    inputs with v = 0 sprinkled in (alone, at tile edges, 70 in a row): is_non_zero stops after its first row
    (/root/reference/src/scalar.rs:69-79), the layout is ragged, the error mask and the prefix sums are compared too."""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    release_hbm()
    v, y, s, a, b = bench.mix_inputs(BATCH, seed=0xC3)
    if form == "bench":
        wl = bench.Workload("c3", engine, DEV, 0, 1, LOG2_BATCH, -1)
        for name in NINE:
            getattr(wl.cols, name).fill_(-1)
        wl.res.fill_(-1)
        wl.launch(0)
        torch.cuda.synchronize()
        cols, res = wl.cols, wl.res
        d_roff, d_voff = wl.plan_buffers[1:]
        zeros = []
    else:
        zeros = sorted(set(range(7, BATCH, 4099)) | set(range(500_000, 500_070)) | {0, 255, 256, 257, BATCH - 1})
        v[zeros] = 0
        ins = [dev(x) for x in (v, y, s, a, b)]
        _, d_roff, d_voff = engine.ragged_buffers(BATCH)
        d_roff.fill_(-1)
        d_voff.fill_(-1)
        err = torch.full((BATCH,), 7, dtype=torch.uint8, device=DEV)
        res = torch.full((BATCH, 2), -1, dtype=torch.int64, device=DEV)
        cols = pg.Columns.allocate(10 * BATCH, 15 * BATCH, DEV, 3, 5)  # the worst case: the layout is not known yet
        for name in NINE:
            getattr(cols, name).fill_(-1)
        engine.scalar_mix_planned(*ins, d_roff, d_voff, cols, res, err, 3, 5, 0)
        torch.cuda.synchronize()
    lay, nerr = engine.plan_result()
    plan = po.scalar_mix_plan(v)
    roff, voff, o_err = plan
    assert (lay.n_gates, lay.n_vars, nerr) == (int(roff[-1]), int(voff[-1]), len(zeros))
    assert torch.equal(d_roff, dev(roff)) and torch.equal(d_voff, dev(voff))
    if form == "failing_items":
        assert torch.equal(err, torch.from_numpy(o_err).to(DEV))
        assert bool((cols.q_m[lay.n_gates:] == -1).all()) and bool((cols.var_values[lay.n_vars:] == -1).all())
    threads = oracle_threads()
    chunk = 1 << 16
    results = []

    def produce(lo, hi, out):
        results.append(po.scalar_mix_fast(v, y, s, a, b, plan, lo, hi, var_base=5, zero_var=0, threads=threads, out=out)["result_vars"])

    def run(only=None):
        return stream_compare(cols, lambda i: (int(roff[i]), int(voff[i])), produce, chunk, chunk * 10, chunk * 15, only_chunk=only)

    n_chunks, words = run()
    assert n_chunks == BATCH // chunk and words * 8 == lay.n_gates * 184 + lay.n_vars * 32
    assert torch.equal(res, dev(np.concatenate(results)))
    results.clear()
    item = 7 * chunk + 12345 if form == "bench" else 500_030  # (the latter: a failing item inside the run of 70)
    k = item // chunk
    flip_and_find(cols, "q_o", int(roff[item]) + 1, 3, lambda: run(only=k))
    flip_and_find(cols, "w_l", int(roff[item]), 0, lambda: run(only=k))
    flip_and_find(cols, "var_values", int(voff[item]) + 5, 1, lambda: run(only=k))
    del cols, run, produce
    release_hbm()


@pytest.mark.parametrize("batch,form", [(524_288 + 77, "planned"), (700_001, "planned"), (700_001, "two_step"), (1_000_003, "two_step")])
def test_fused_mix_early_rows_every_limb(engine, batch, form):
    """the fused mix at sizes where the arithmetic launch writes row tiles itself while it inverts (calls of half a million items and
    more: csrc/scalar_gadgets.hpp, "early rows"), with everything that decides which tiles those are: a batch that is no multiple of
    a workgroup's span or of a tile, failing items in some spans and not in others (first, last and middle spans; a span whose only
    failing item is its last), wire columns on an odd 8-byte boundary, the planned call (prefix sums by look-back: the writers wait for
    them) and plan + emit.  Every row and variable against the oracle; nothing beyond the totals touched."""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    release_hbm()
    v, y, s, a, b = bench.mix_inputs(batch, seed=batch & 0xFFFF)
    ipl = -(-batch // (256 * 8 * 32))
    span = 256 * ipl
    zeros = sorted({0, span + 5, 3 * span - 1, 7 * span, 7 * span + 1, batch - 1} | set(range(10 * span + 300, 10 * span + 330)))
    zeros = [z for z in zeros if z < batch]
    v[zeros] = 0
    ins = [dev(x) for x in (v, y, s, a, b)]
    plan = po.scalar_mix_plan(v)
    roff, voff, o_err = plan
    G, V = int(roff[-1]), int(voff[-1])
    _, d_roff, d_voff = engine.ragged_buffers(batch)
    big = pg.Columns.allocate(10 * batch + 2, 15 * batch + 2, DEV)
    for name in NINE:
        getattr(big, name).fill_(-1)
    # wire columns one element in: an odd 8-byte boundary, as a composer appending at an odd row has them
    cols = pg.Columns(big.q_m[:10 * batch], big.q_l[:10 * batch], big.q_r[:10 * batch], big.q_o[:10 * batch], big.q_c[:10 * batch],
                      big.w_l[1:1 + 10 * batch], big.w_r[1:1 + 10 * batch], big.w_o[1:1 + 10 * batch], big.var_values[:15 * batch])
    res = torch.full((batch, 2), -1, dtype=torch.int64, device=DEV)
    if form == "planned":
        err = torch.full((batch,), 7, dtype=torch.uint8, device=DEV)
        engine.scalar_mix_planned(*ins, d_roff, d_voff, cols, res, err, 3, 5, 0)
        torch.cuda.synchronize()
        assert torch.equal(err, torch.from_numpy(o_err).to(DEV))
    else:
        lay0, nerr0 = engine.scalar_mix_plan(ins[0], d_roff, d_voff)
        assert (lay0.n_gates, lay0.n_vars, nerr0) == (G, V, len(zeros))
        engine.scalar_mix_emit(*ins, d_roff, d_voff, cols, res, 3, 5, 0)
        torch.cuda.synchronize()
    lay, nerr = engine.plan_result()
    assert (lay.n_gates, lay.n_vars, nerr) == (G, V, len(zeros))
    assert torch.equal(d_roff, dev(roff)) and torch.equal(d_voff, dev(voff))
    threads = oracle_threads()
    chunk = 1 << 16
    results = []

    def produce(lo, hi, out):
        results.append(po.scalar_mix_fast(v, y, s, a, b, plan, lo, hi, var_base=5, zero_var=0, threads=threads, out=out)["result_vars"])

    n_chunks, words = stream_compare(cols, lambda i: (int(roff[i]), int(voff[i])), produce, chunk, chunk * 10, chunk * 15, n_items=batch)
    assert words * 8 == G * 184 + V * 32
    assert torch.equal(res, dev(np.concatenate(results)))
    for name in SEL:
        assert bool((getattr(big, name)[G:] == -1).all()), name
    for name in WIRES:
        t = getattr(big, name)
        assert int(t[0]) == -1 and bool((t[1 + G:] == -1).all()), name
    assert bool((big.var_values[V:] == -1).all())
    del big, cols, produce
    release_hbm()


def test_config_c2_witness_refresh_every_limb(engine):
    """the witness refresh of config 2's circuit (pg_range_check_values_batch: the assignments alone, the reference's prover flow after
    clear_witness(), /root/reference/tests/scalar_gadgets_tests.rs:108-119) at full size through bench.Workload's launch: all
    1 084 227 584 variables of the table, 34.7 GB, limb for limb against the oracle's var_values for the same witnesses -- into a table
    that held OTHER witnesses' assignments before, with a guard behind it"""
    import bench
    import types
    from oracle import pyoracle as po
    G, V = 1031, 1034
    release_hbm()
    wl = bench.Workload("c2_values", engine, DEV, 0, 1, LOG2_BATCH, -1)
    table = wl.cols
    assert table.shape == (BATCH * V, 4)
    guard = torch.full((64, 4), -1, dtype=torch.int64, device=DEV)
    table.fill_(0x3C3C3C3C3C3C3C3C)
    wl.launch(0)
    torch.cuda.synchronize()
    wit = synth.random_scalars(BATCH, seed=synth.SEED + 1)  # (bench.Workload's witnesses for this workload)
    mn, mx = synth.mont(0), synth.mont(2**254)
    threads = oracle_threads()
    chunk = 1 << 12
    cols = types.SimpleNamespace(var_values=table)

    def produce(lo, hi, out):
        po.range_check_fast(mn, mx, np.ascontiguousarray(wit[lo:hi]), threads=threads, var_base=5 + lo * V, out=out)

    def run(only=None):
        return stream_compare(cols, lambda i: (i * G, i * V), produce, chunk, chunk * G, chunk * V, only_chunk=only, names=("var_values",))

    n_chunks, words = run()
    assert n_chunks == BATCH // chunk and words * 8 == BATCH * V * 32 == 34_695_282_688
    assert bool((guard == -1).all())
    item = 200 * chunk + 77
    flip_and_find(cols, "var_values", item * V + 1033, 0, lambda: run(only=200))   # the item's last variable: the outcome
    flip_and_find(cols, "var_values", item * V + 259 + 255 + 1, 2, lambda: run(only=200))  # z of the first bound block
    wl.release()
    del wl, table, cols, run, produce
    release_hbm()


def test_witness_refresh_beyond_2_pow_32_units(engine):
    """ONE pg_range_check_values_batch call over 2^22 witnesses: 4 336 910 336 Variables = 8.67e9 sixteen-byte units, 138.8 GB -- a
    call whose unit count (and every byte offset past the first quarter) does not fit 32 bits, where the sweeps' tile-relative 32-bit
    counters feed 64-bit global offsets (csrc/emit.hpp).  Every limb of the table against the threaded oracle
    (/root/reference/src/range.rs:119-158: the ladder's assignments), guard regions before and after the table."""
    from oracle import pyoracle as po
    import plonk_gadgets_amd as pg
    import types
    G, V, LOG2 = 1031, 1034, 22
    batch = 1 << LOG2
    release_hbm()
    free, _ = torch.cuda.mem_get_info()
    if free < batch * V * 32 + (16 << 30):
        pytest.skip("not enough free HBM for a 139-GB table")
    GUARD = 4096
    big = torch.empty((batch * V + 2 * GUARD, 4), dtype=torch.int64, device=DEV)
    big[:GUARD].fill_(-1)
    big[GUARD + batch * V:].fill_(-1)
    table = big[GUARD:GUARD + batch * V]
    table.fill_(0x3C3C3C3C3C3C3C3C)
    wit = synth.random_scalars(batch, seed=synth.SEED + 22)
    mn, mx = synth.mont(0), synth.mont(2**254)
    engine.range_check_values_batch(pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254), dev(wit), table)
    torch.cuda.synchronize()
    assert bool((big[:GUARD] == -1).all()) and bool((big[GUARD + batch * V:] == -1).all())
    threads = oracle_threads()
    chunk = 1 << 13
    cols = types.SimpleNamespace(var_values=table)

    def produce(lo, hi, out):
        po.range_check_fast(mn, mx, np.ascontiguousarray(wit[lo:hi]), threads=threads, var_base=5 + lo * V,
                            out={"var_values": out["var_values"]})   # (the assignments alone: no rows are made)

    def run(only=None):
        return stream_compare(cols, lambda i: (0, i * V), produce, chunk, 1, chunk * V, n_items=batch, only_chunk=only, names=("var_values",))

    n_chunks, words = run()
    assert n_chunks == batch // chunk and words * 8 == batch * V * 32 == 138_781_130_752
    last = batch - 1                                       # (the very last Variable: unit 8.67e9)
    flip_and_find(cols, "var_values", last * V + 1033, 3, lambda: run(only=n_chunks - 1))
    item = (1 << 21) + 5                                   # (just past 2^32 units: Variable 2^31 + ...)
    flip_and_find(cols, "var_values", item * V + 2, 0, lambda: run(only=item // chunk))
    del big, table, cols, run, produce
    release_hbm()
