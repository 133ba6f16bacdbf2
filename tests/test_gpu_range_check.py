"""GPU parity: HIP range_check path (through the C ABI) vs the CPU oracle, limb for limb."""
import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

SCALAR_COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "var_values")
WIRE_COLS = ("w_l", "w_r", "w_o")
Q = synth.Q


@pytest.fixture(scope="module")
def engine():
    import plonk_gadgets_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


def run_gpu(engine, mn, mx, wit_np, gate_base=3, var_base=5):
    import plonk_gadgets_amd as pg
    w = torch.from_numpy(wit_np.view(np.int64)).to("cuda:0")
    cols, res = engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), w, gate_base, var_base)
    torch.cuda.synchronize()
    out = cols.to_numpy()
    out["result_vars"] = res.cpu().numpy().view(np.uint64)
    return out


def assert_same(gpu, ora):
    for k in SCALAR_COLS + WIRE_COLS + ("result_vars",):
        a, b = gpu[k], ora[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)[0]
            raise AssertionError(f"{k} differs first at {bad.tolist()}: gpu={a[tuple(bad)]:#x} oracle={b[tuple(bad)]:#x}")


def mixed_witnesses(mn, mx, n, seed):
    """about half in range, half out, plus the edges"""
    span = max(mx - mn, 1)
    inside = [mn + int(v) % span for v in synth.splitmix64(n, seed)]
    outside = synth.random_scalars(n, seed + 1)
    edges = [mn, mx - 1, mx, (mn - 1) % Q, 0, Q - 1]
    arr = np.concatenate([synth.scalars_from_ints(inside), outside, synth.scalars_from_ints(edges)])
    perm = np.argsort(synth.splitmix64(len(arr), seed + 2), kind="stable")
    return np.ascontiguousarray(arr[perm])


CONFIGS = [
    (0, 2**64, 21),                    # BASELINE config C1 shape: n = 65
    (50_000, 250_000, 9),              # the reference's range_check cases: n = 19
    (2**126, 2**127 + 1, 5),           # reference case 7: n = 129
    (0, 2**254, 40),                   # BASELINE config C2 shape: n = 255
    (0, 2, 3),                         # n = 2 (smallest ladder)
    (7, 2**253 + 12345, 6),            # n = 255 via bitlen 254
    (1, Q - 1, 4),                     # max-1 has 255 bits -> n = 252
]


@pytest.mark.parametrize("mn,mx,count", CONFIGS)
def test_range_check_matches_oracle(engine, mn, mx, count):
    from oracle import pyoracle as po
    wit = mixed_witnesses(mn, mx, count, seed=mx % 1000003)
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), wit)
    assert ora["satisfied"] and (ora["gate_base"], ora["var_base"]) == (3, 5)
    gpu = run_gpu(engine, mn, mx, wit)
    assert_same(gpu, ora)


@pytest.mark.parametrize("batch", [1, 2, 15, 16, 17, 31, 33])
def test_ragged_tile_sizes(engine, batch):
    """batch sizes around the 16-item tile; odd gate_base/var_base so wire columns start 8-byte-odd"""
    from oracle import pyoracle as po
    mn, mx = 50_000, 250_000
    wit = mixed_witnesses(mn, mx, batch, seed=batch)[:batch]
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), wit)
    gpu = run_gpu(engine, mn, mx, wit)
    assert_same(gpu, ora)


def test_base_offsets_are_parameters(engine):
    """rows do not depend on gate_base; wires/result vars shift with var_base (composer state is a parameter)"""
    mn, mx = 0, 2**64
    wit = mixed_witnesses(mn, mx, 10, seed=99)
    a = run_gpu(engine, mn, mx, wit, gate_base=3, var_base=5)
    b = run_gpu(engine, mn, mx, wit, gate_base=1000, var_base=123456789012)
    for k in SCALAR_COLS:
        assert np.array_equal(a[k], b[k])
    for k in WIRE_COLS + ("result_vars",):
        assert np.array_equal(a[k] - np.uint64(5), b[k] - np.uint64(123456789012))


def test_empty_batch(engine):
    import plonk_gadgets_amd as pg
    w = torch.empty((0, 4), dtype=torch.int64, device="cuda:0")
    cols, res = engine.range_check_batch(pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**64), w)
    assert cols.q_m.shape[0] == 0 and res.shape[0] == 0


def test_structure_is_witness_independent(engine):
    """tests/scalar_gadgets_tests.rs:36 vs :43 -- verifier builds the same circuit from other witnesses"""
    mn, mx = 50_000, 250_000
    a = run_gpu(engine, mn, mx, mixed_witnesses(mn, mx, 12, seed=1))
    b = run_gpu(engine, mn, mx, mixed_witnesses(mn, mx, 12, seed=2))
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c") + WIRE_COLS:
        assert np.array_equal(a[k], b[k])


def test_large_batch_properties_and_samples(engine):
    """2^14 items at n = 255 (3.6 GB of columns): periodic selectors, affine wires, sampled items vs the oracle"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    mn, mx, batch = 0, 2**254, 1 << 14
    wit = synth.random_scalars(batch, seed=20260101)
    w = torch.from_numpy(wit.view(np.int64)).to("cuda:0")
    cols, res = engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), w, 3, 5)
    torch.cuda.synchronize()
    lay = engine.range_check_layout(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), batch)
    G, V = lay.gates_per_item, lay.vars_per_item
    assert (G, V) == (1031, 1034)
    for name in ("q_m", "q_l", "q_r", "q_o", "q_c"):
        c = getattr(cols, name).view(batch, G, 4)
        assert bool((c == c[0:1]).all()), name
    item_base = (torch.arange(batch, device="cuda:0", dtype=torch.int64) * V).view(batch, 1)
    for name in WIRE_COLS:
        c = getattr(cols, name).view(batch, G) - item_base
        assert bool((c == c[0:1]).all()), name
    assert bool((res == 5 + item_base.view(-1) + V - 1).all())
    # the witness itself is the first variable of every item
    assert bool((cols.var_values.view(batch, V, 4)[:, 0, :] == w).all())
    # sampled items, limb for limb (wire indices relocated to the sample's own numbering)
    idx = [0, 1, 15, 16, 17, 4095, 8191, batch - 1] + [int(x) % batch for x in synth.splitmix64(8, 5)]
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), np.ascontiguousarray(wit[idx]))
    for s, i in enumerate(idx):
        for name in ("q_m", "q_l", "q_r", "q_o", "q_c"):
            got = getattr(cols, name)[i * G:(i + 1) * G].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, ora[name][s * G:(s + 1) * G]), (name, i)
        got = cols.var_values[i * V:(i + 1) * V].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ora["var_values"][s * V:(s + 1) * V]), ("var_values", i)
        for name in WIRE_COLS:
            got = getattr(cols, name)[i * G:(i + 1) * G].cpu().numpy().view(np.uint64) - np.uint64(i * V)
            assert np.array_equal(got, ora[name][s * G:(s + 1) * G] - np.uint64(s * V)), (name, i)


@pytest.mark.parametrize("name", ["range_check_ref_50k_250k", "range_check_ref_2p126_2p127", "range_check_c1_n65",
                                  "range_check_c2_n255"])
def test_range_check_golden_fixtures(engine, name):
    """HIP path vs the committed golden vectors (tests/golden/, inputs = the reference's own test cases)"""
    import os
    import plonk_gadgets_amd as pg
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    w = torch.from_numpy(g["witness"].view(np.int64)).to("cuda:0")
    cols, res = engine.range_check_batch(pg.BlsScalar.from_limbs(g["min_range"][0]),
                                         pg.BlsScalar.from_limbs(g["max_range"][0]), w, 3, 5)
    torch.cuda.synchronize()
    got = cols.to_numpy()
    for k in SCALAR_COLS + WIRE_COLS:
        assert np.array_equal(got[k], g[k]), k
    assert np.array_equal(res.cpu().numpy().view(np.uint64), g["result_vars"])


def test_misaligned_wire_columns(engine):
    """wire columns that start on an odd 8-byte boundary (a composer appending at an odd gate index)"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    mn, mx = 50_000, 250_000
    wit = mixed_witnesses(mn, mx, 9, seed=77)
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), wit)
    lay = engine.range_check_layout(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), len(wit))
    big = pg.Columns.allocate(lay.n_gates + 2, lay.n_vars + 2, "cuda:0")
    for t in (big.w_l, big.w_r, big.w_o):
        t.fill_(-1)
    view = pg.Columns(big.q_m[1:-1], big.q_l[1:-1], big.q_r[1:-1], big.q_o[1:-1], big.q_c[1:-1],
                      big.w_l[1:-1], big.w_r[2:], big.w_o[1:-1], big.var_values[1:-1])
    w = torch.from_numpy(wit.view(np.int64)).to("cuda:0")
    engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), w, 3, 5, out=view)
    torch.cuda.synchronize()
    got = view.to_numpy()
    for k in SCALAR_COLS + WIRE_COLS:
        assert np.array_equal(got[k], ora[k]), k
    # guard elements around the odd-aligned columns untouched
    assert int(big.w_l[0]) == -1 and int(big.w_l[-1]) == -1 and int(big.w_o[0]) == -1 and int(big.w_o[-1]) == -1
    assert int(big.w_r[0]) == -1 and int(big.w_r[1]) == -1


def test_columns_in_one_spread_slab(engine):
    """Columns.allocate(spread_gib=...): nine views of ONE allocation, the selector columns a stride apart -- disjoint, inside
    the slab, 16-byte aligned; a call that writes into them == the oracle, and the slab between the arrays keeps its fill"""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    mn, mx = 0, 2**64
    wit = mixed_witnesses(mn, mx, 21, seed=91)
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), wit)
    lay = engine.range_check_layout(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), len(wit))
    cols = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0", spread_gib=0.03)
    names = SCALAR_COLS[:5] + WIRE_COLS + ("var_values",)
    lo, hi = cols.slab.data_ptr(), cols.slab.data_ptr() + cols.slab.numel() * 8
    iv = sorted((getattr(cols, n).data_ptr(), getattr(cols, n).data_ptr() + getattr(cols, n).numel() * 8) for n in names)
    assert iv[0][0] >= lo and iv[-1][1] <= hi and all(a1 <= b0 for (_, a1), (b0, _) in zip(iv, iv[1:]))
    assert all(getattr(cols, n).data_ptr() % 16 == 0 for n in names)
    sel = sorted(getattr(cols, n).data_ptr() for n in SCALAR_COLS[:5])
    assert all(b - a >= int(0.03 * 2**30) - (2 << 20) for a, b in zip(sel, sel[1:]))
    cols.slab.fill_(0x5A5A5A5A5A5A5A5A)
    w = torch.from_numpy(wit.view(np.int64)).to("cuda:0")
    engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), w, 3, 5, out=cols)
    torch.cuda.synchronize()
    got = cols.to_numpy()
    for k in SCALAR_COLS + WIRE_COLS:
        assert np.array_equal(got[k], ora[k]), k
    mask = torch.ones_like(cols.slab, dtype=torch.bool)
    for a, b in iv:
        mask[(a - lo) // 8:(b - lo) // 8] = False
    assert bool((cols.slab[mask] == 0x5A5A5A5A5A5A5A5A).all())


def test_config_c1_exact(engine):
    """BASELINE config 1: 1 000 x range_check(v, "64-bit") -- min = 0, max = 2^64 (n = 65, 271 rows per witness),
    witnesses uniform in [0, 2^64 + 2^60) (about 6 % out of range): every limb vs the CPU oracle."""
    from oracle import pyoracle as po
    mn, mx = 0, 2**64
    wit = synth.uniform_below(1000, 2**64 + 2**60, seed=synth.SEED)
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), wit)
    assert ora["satisfied"] and ora["n_gates"] == 271_000 and ora["n_vars"] == 654_000
    gpu = run_gpu(engine, mn, mx, wit)
    assert_same(gpu, ora)
    outcomes = [synth.to_int(gpu["var_values"][int(r) - 5]) for r in gpu["result_vars"]]
    assert outcomes == [int(synth.to_int(w) < 2**64) for w in wit] and 0 < sum(outcomes) < 1000


def test_config_c2_full_size_properties(engine):
    """BASELINE config 2 at its full size: 2^20 witnesses x n = 255 in ONE launch (233.6 GB of columns).
    Size-independent properties on the device (periodic selectors, affine wires, witness = first variable,
    result variables) + items sampled across the whole range vs the CPU oracle."""
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    free, _ = torch.cuda.mem_get_info()
    batch = 1 << 20
    G, V = 1031, 1034
    if free < batch * (G * 184 + V * 32) + (24 << 30):
        pytest.skip("not enough free HBM for the full-size batch")
    mn, mx = 0, 2**254
    wit = synth.random_scalars(batch, seed=synth.SEED)
    w = torch.from_numpy(wit.view(np.int64)).to("cuda:0")
    # every output array 64 entries longer than the layout and filled with a sentinel: a slot nobody writes fails the row
    # check and the comparisons below, a store beyond the layout shows in the tails
    GUARD = 64
    big = pg.Columns.allocate(batch * G + GUARD, batch * V + GUARD, "cuda:0", 3, 5)
    for name in SCALAR_COLS + WIRE_COLS:
        getattr(big, name).fill_(-1)
    cols = pg.Columns(**{n: getattr(big, n)[:batch * G] for n in SCALAR_COLS[:-1] + WIRE_COLS}, var_values=big.var_values[:batch * V],
                      gate_base=3, var_base=5)
    res = torch.full((batch + GUARD,), -1, dtype=torch.int64, device="cuda:0")
    engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), w, 3, 5, out=cols, result_vars=res[:batch])
    torch.cuda.synchronize()
    for name in SCALAR_COLS[:-1] + WIRE_COLS:
        assert bool((getattr(big, name)[batch * G:] == -1).all()), name
    assert bool((big.var_values[batch * V:] == -1).all()) and bool((res[batch:] == -1).all())
    res = res[:batch]
    step = 1 << 14  # compare in slabs to bound temporary memory
    for name in ("q_m", "q_l", "q_r", "q_o", "q_c"):
        c = getattr(cols, name).view(batch, G, 4)
        ref = c[0:1]
        for s in range(0, batch, step):
            assert bool((c[s:s + step] == ref).all()), (name, s)
    for name in WIRE_COLS:
        c = getattr(cols, name).view(batch, G)
        ref = c[0:1]
        for s in range(0, batch, step):
            base = (torch.arange(s, min(s + step, batch), device="cuda:0", dtype=torch.int64) * V).view(-1, 1)
            assert bool(((c[s:s + step] - base) == ref).all()), (name, s)
    assert bool((res == 5 + torch.arange(batch, device="cuda:0", dtype=torch.int64) * V + (V - 1)).all())
    vv = cols.var_values.view(batch, V, 4)
    assert bool((vv[:, 0, :] == w).all())
    # every result is 1 at n = 255 (every field element fits 255 bits): the last variable of each item is mont(1)
    one = torch.tensor(np.array(synth.mont(1), dtype=np.uint64).view(np.int64), device="cuda:0")
    assert bool((vv[:, V - 1, :] == one).all())
    # every one of the 1.08e9 rows satisfies its gate equation over the emitted variable table (device-side check)
    assert engine.check_rows(cols, var_base=5) == -1
    idx = [0, 31, 32, 33, batch // 2 - 1, batch // 2, batch - 33, batch - 1] + [int(x) % batch for x in synth.splitmix64(8, 9)]
    ora = po.range_check_batch(synth.mont(mn), synth.mont(mx), np.ascontiguousarray(wit[idx]))
    for s, i in enumerate(idx):
        for name in ("q_m", "q_l", "q_r", "q_o", "q_c"):
            got = getattr(cols, name)[i * G:(i + 1) * G].cpu().numpy().view(np.uint64)
            assert np.array_equal(got, ora[name][s * G:(s + 1) * G]), (name, i)
        got = cols.var_values[i * V:(i + 1) * V].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ora["var_values"][s * V:(s + 1) * V]), ("var_values", i)
        for name in WIRE_COLS:
            got = getattr(cols, name)[i * G:(i + 1) * G].cpu().numpy().view(np.uint64) - np.uint64(i * V)
            assert np.array_equal(got, ora[name][s * G:(s + 1) * G] - np.uint64(s * V)), (name, i)
    del cols, vv, big
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mn,mx,batch", [(0, 2**254, 4096), (2**126, 2**127 + 1, 6000), (0, 2**64, 20000)])
def test_medium_batches_every_limb(engine, mn, mx, batch):
    """thousands of items, every limb of every column, vs oracle/fast.c (itself pinned to the faithful restatement
    by tests/test_oracle_fast.py): about half of the witnesses out of range for the n = 129 and n = 65 shapes"""
    from oracle import pyoracle as po
    wit = mixed_witnesses(mn, mx, batch // 2, seed=batch)[:batch]
    ora = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit, threads=8, var_base=5)
    gpu = run_gpu(engine, mn, mx, wit)
    assert_same(gpu, ora)


def test_fuzz_ladder_lengths(engine):
    """48 random (min, max) pairs covering ladder lengths all over 2..255, random batch sizes around the tile width,
    witnesses inside / outside / at the edges: every limb vs oracle/fast.c"""
    from oracle import pyoracle as po
    import random
    rng = random.Random(20261003)
    seen = set()
    for trial in range(48):
        bits = rng.choice([1, 2, 3, 7, 8, 31, 32, 33, 63, 64, 65, 127, 128, 200, 251, 252, 253, 254, 255]) if trial < 19 \
            else rng.randrange(1, 256)
        mx = rng.randrange(1 << (bits - 1), min(1 << bits, Q)) if bits > 1 else rng.choice([1, 2])
        mn = rng.randrange(0, mx) if rng.random() < 0.7 else 0
        batch = rng.choice([1, 2, 31, 32, 33, 63, 64, 65]) if trial % 3 == 0 else rng.randrange(1, 80)
        ws = []
        for i in range(batch):
            c = rng.random()
            ws.append(rng.randrange(mn, mx) if c < 0.45 and mx > mn else rng.randrange(Q) if c < 0.8
                      else rng.choice([mn, mx - 1, mx, (mn - 1) % Q, 0, Q - 1]))
        wit = synth.scalars_from_ints(ws)
        ora = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit, threads=4, var_base=5)
        seen.add(ora["num_bits"])
        gpu = run_gpu(engine, mn, mx, wit)
        try:
            assert_same(gpu, ora)
        except AssertionError as e:
            raise AssertionError(f"trial {trial}: min={mn:#x} max={mx:#x} n={ora['num_bits']} batch={batch}: {e}")
    assert len(seen) >= 25 and {2, 252, 255} & seen


def test_structure_only_rows(engine):
    """pg_range_check_structure_batch: the selectors and wire indices of the full call, bit for bit, from the public
    bounds and the numbering alone; the variable table is not touched"""
    import plonk_gadgets_amd as pg
    for mn, mx, batch in ((50_000, 250_000, 77), (0, 2**254, 40)):
        wit = torch.from_numpy(synth.random_scalars(batch, 3).view(np.int64)).to("cuda:0")
        full, _ = engine.range_check_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), wit, 1234, 56789)
        lay = engine.range_check_layout(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), batch)
        only = pg.Columns.allocate(lay.n_gates, lay.n_vars, "cuda:0")
        only.var_values.fill_(-7)
        engine.range_check_structure_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), batch, 1234, 56789, only)
        torch.cuda.synchronize()
        for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o"):
            assert torch.equal(getattr(full, k), getattr(only, k)), k
        assert bool((only.var_values == -7).all())
        rows_only = pg.Columns.allocate(lay.n_gates, 0, "cuda:0")          # no variable table at all
        engine.range_check_structure_batch(pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), batch, 1234, 56789, rows_only)
        assert torch.equal(rows_only.w_o, full.w_o) and torch.equal(rows_only.q_c, full.q_c)


def test_host_pipeline(engine):
    """chunks streamed to pinned host memory while the next chunk is emitted: every chunk the consumer sees equals the
    oracle's rows for those items at their global numbering"""
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd.host_pipeline import HostPipeline
    from oracle import pyoracle as po
    mn, mx, total, chunk = 50_000, 250_000, 96, 16
    wit = np.ascontiguousarray(mixed_witnesses(mn, mx, total, 5)[:total])
    ora = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit, threads=2, var_base=5)
    G, V = 4 * ora["num_bits"] + 11, 2 * ora["num_bits"] + 524
    pipe = HostPipeline(engine, pg.BlsScalar.from_int(mn), pg.BlsScalar.from_int(mx), chunk)
    seen = []

    def consume(cols, k, first):
        assert first == k * chunk
        for name in ("q_m", "q_l", "q_r", "q_o", "q_c"):
            assert np.array_equal(getattr(cols, name).numpy().view(np.uint64), ora[name][first * G:(first + chunk) * G]), (k, name)
        for name in ("w_l", "w_r", "w_o"):
            assert np.array_equal(getattr(cols, name).numpy().view(np.uint64), ora[name][first * G:(first + chunk) * G]), (k, name)
        assert np.array_equal(cols.var_values.numpy().view(np.uint64), ora["var_values"][first * V:(first + chunk) * V]), k
        seen.append(k)

    pipe.run(torch.from_numpy(wit.view(np.int64)).to("cuda:0"), 3, 5, consume=consume)
    assert seen == list(range(total // chunk))
