"""GPU: the multi-GPU slice of the path on the REAL engine (one rank: every box of the pool has one GPU).

tests/test_distributed_cpu.py covers the N > 1 arithmetic with gloo and a stand-in engine; here the same pipelines
drive pg.Engine -- emission on the compute stream, the inversion pre-pass on the engine's side stream, the collective
on the communicator's stream -- and a consumer compares every chunk with the oracle as it arrives, so an ordering bug
between the three streams shows up as a wrong limb.  The collective is the library's own (pg_comm: RCCL bound at run
time, communicator of size 1) and, for comparison, the world = 1 form of the torch one."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from plonk_gadgets_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")
MN, MX = 50_000, 250_000


@pytest.fixture(scope="module")
def engine():
    import plonk_gadgets_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


def witnesses(total, seed=11):
    inside = synth.scalars_from_ints([MN + int(v) % (MX - MN) for v in synth.splitmix64(total // 2, seed)])
    return np.ascontiguousarray(np.concatenate([inside, synth.random_scalars(total - total // 2, seed + 1)]))


def oracle_rows(wit):
    from oracle import pyoracle as po
    ora = po.range_check_batch(synth.mont(MN), synth.mont(MX), wit)  # fresh composer: rows from gate 3, variables from 5
    assert ora["satisfied"] and (ora["gate_base"], ora["var_base"]) == (3, 5)
    return ora, 4 * ora["num_bits"] + 11, 2 * ora["num_bits"] + 524


def expect(ora, name, first, count, G, V):
    if name == "var_values":
        return ora[name][first * V:(first + count) * V]
    return ora[name][first * G:(first + count) * G]


@pytest.mark.parametrize("native", [True, False])
def test_gather_pipeline_on_the_engine(engine, native):
    """GatherPipeline with pg.Engine: 6 chunks through the double buffer, every gathered chunk == the oracle's rows of
    those items at their global numbering (half of the witnesses fail the range: their inverses come from the pre-pass
    on the side stream and must have landed before the collective reads the chunk)"""
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import distributed as pd
    total, chunk = 96, 16
    wit = witnesses(total)
    ora, G, V = oracle_rows(wit)
    mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)
    coll = pd.NativeCollective(engine) if native else pd.TorchCollective()
    pipe = pd.GatherPipeline(engine, mn, mx, chunk, collective=coll)
    assert pipe.world == 1 and pipe.rank == 0
    seen = []

    def consume(gathered, k):
        assert gathered.shape == (1, pipe.words)
        part = pd.columns_in(gathered[0], pipe.lay.n_gates, pipe.lay.n_vars)
        for name in COLS:
            got = getattr(part, name).cpu().numpy().view(np.uint64)
            assert np.array_equal(got, expect(ora, name, k * chunk, chunk, G, V)), (k, name)
        seen.append(k)

    pipe.run(torch.from_numpy(wit.view(np.int64)).to("cuda:0"), total, 3, 5, consume=consume)
    assert seen == list(range(total // chunk))
    # a second run through the same buffers, without a consumer in between the chunks (no host synchronisation at all
    # until the end): the last two chunks must still be intact
    last = []
    pipe.run(torch.from_numpy(wit.view(np.int64)).to("cuda:0"), total, 3, 5,
             consume=lambda g, k: last.append((k, g.clone())) if k >= total // chunk - 2 else None)
    torch.cuda.synchronize()
    for k, g in last:
        part = pd.columns_in(g[0], pipe.lay.n_gates, pipe.lay.n_vars)
        for name in COLS:
            assert np.array_equal(getattr(part, name).cpu().numpy().view(np.uint64), expect(ora, name, k * chunk, chunk, G, V)), (k, name)
    coll.close()


@pytest.mark.parametrize("native", [True, False])
def test_variables_only_pipeline_on_the_engine(engine, native):
    """VariablesOnlyPipeline with pg.Engine: only the variable tables go through the collective; every part the consumer
    sees is complete and equal to the oracle"""
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import distributed as pd
    total, chunk = 80, 16
    wit = witnesses(total, seed=23)
    ora, G, V = oracle_rows(wit)
    mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)
    coll = pd.NativeCollective(engine) if native else pd.TorchCollective()
    pipe = pd.VariablesOnlyPipeline(engine, mn, mx, chunk, collective=coll)
    seen = []

    def consume(parts, k):
        assert len(parts) == 1
        for name in COLS:
            got = getattr(parts[0], name).cpu().numpy().view(np.uint64)
            assert np.array_equal(got, expect(ora, name, k * chunk, chunk, G, V)), (k, name)
        seen.append(k)

    pipe.run(torch.from_numpy(wit.view(np.int64)).to("cuda:0"), total, 3, 5, consume=consume)
    assert seen == list(range(total // chunk))
    assert pipe.bytes_on_the_links_per_chunk() == chunk * V * 32
    coll.close()


def test_sharded_batch_through_the_c_abi(engine):
    """pg_range_check_sharded_batch for rank 0, 1, 2 of 3 (uneven shards: 4 + 3 + 3), each into its slice of ONE set of
    columns: together they are the single-process loop over all ten witnesses, limb for limb"""
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import distributed as pd
    total, world = 10, 3
    wit = witnesses(total, seed=31)
    ora, G, V = oracle_rows(wit)
    mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)
    dw = torch.from_numpy(wit.view(np.int64)).to("cuda:0")
    full = pg.Columns.allocate(total * G, total * V, "cuda:0")
    res = torch.empty((total,), dtype=torch.int64, device="cuda:0")
    for rank in range(world):
        lo, hi = pd.shard_range(total, rank, world)
        info = pd.range_check_shard_layout(mn, mx, total, rank, world, 3, 5)
        assert (info.lo, info.hi, info.gate_base, info.var_base) == (lo, hi, 3 + lo * G, 5 + lo * V)
        part = pg.Columns(*[getattr(full, n)[lo * G:hi * G] for n in COLS[:-1]], full.var_values[lo * V:hi * V])
        engine.range_check_sharded_batch(mn, mx, dw[lo:hi], total, rank, world, 3, 5, out=part, result_vars=res[lo:hi])
    torch.cuda.synchronize()
    got = full.to_numpy()
    for name in COLS:
        assert np.array_equal(got[name], ora[name]), name
    assert np.array_equal(res.cpu().numpy().view(np.uint64), ora["result_vars"])


def test_native_communicator_of_size_one(engine):
    """pg_comm_unique_id / pg_comm_create / pg_allgather_bytes / pg_allgather_columns through ctypes: RCCL is found in
    the process at run time, a communicator of one rank gathers a packed chunk and the nine columns onto themselves"""
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib, distributed as pd
    lib = _lib.load()
    ident = (C.c_uint8 * 128)()
    assert lib.pg_comm_unique_id(ident) == 0, lib.pg_last_error()
    comm = C.c_void_p()
    assert lib.pg_comm_create(engine._h, ident, 1, 1, C.byref(comm)) == 2  # rank >= world
    assert lib.pg_comm_create(engine._h, ident, 0, 1, C.byref(comm)) == 0, lib.pg_last_error()
    assert lib.pg_comm_world(comm) == 1 and lib.pg_comm_rank(comm) == 0
    batch = 24
    wit = witnesses(batch, seed=41)
    mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)
    cols, _ = engine.range_check_batch(mn, mx, torch.from_numpy(wit.view(np.int64)).to("cuda:0"), 3, 5)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # the nine columns, one grouped launch
    gathered = pg.Columns.allocate(cols.q_m.shape[0], cols.var_values.shape[0], "cuda:0")
    for n in COLS:
        getattr(gathered, n).fill_(-1)
    lc, gc = cols.as_c(), gathered.as_c()
    assert lib.pg_allgather_columns(comm, C.byref(lc), cols.q_m.shape[0], cols.var_values.shape[0], C.byref(gc), st) == 0, lib.pg_last_error()
    # raw bytes
    src = torch.arange(4096, dtype=torch.int64, device="cuda:0")
    dst = torch.zeros_like(src)
    assert lib.pg_allgather_bytes(comm, src.data_ptr(), dst.data_ptr(), src.numel() * 8, st) == 0, lib.pg_last_error()
    assert lib.pg_allgather_bytes(comm, src.data_ptr(), dst.data_ptr(), 12, st) == 2  # not a multiple of 8
    assert lib.pg_allgather_bytes(None, src.data_ptr(), dst.data_ptr(), 8, st) == 2
    torch.cuda.synchronize()
    for n in COLS:
        assert torch.equal(getattr(gathered, n), getattr(cols, n)), n
    assert torch.equal(src, dst)
    # the Python layer's one-launch gather of equal shards
    coll = pd.NativeCollective(engine)
    full, res = pd.gather_columns(cols, None, [cols.q_m.shape[0]], [cols.var_values.shape[0]], collective=coll)
    torch.cuda.synchronize()
    for n in COLS:
        assert torch.equal(getattr(full, n), getattr(cols, n)), n
    coll.close()
    lib.pg_comm_destroy(comm)


def test_ragged_sharded_batch_through_the_c_abi(engine):
    """pg_max_bound_ragged_sharded_plan / _batch with a communicator of one rank: plan, exchange of the totals, emission at
    the numbering the exchange gives -- the same columns as the oracle's; and the Python layer's max_bound_ragged_sharded
    through the same entry point"""
    import bench
    import plonk_gadgets_amd as pg
    from oracle import pyoracle as po
    from plonk_gadgets_amd import _lib, distributed as pd
    lib = _lib.load()
    batch = 300
    mr_np, wt_np = bench.c4_inputs(batch, seed=0xC4)
    mr, wt = (torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to("cuda:0") for x in (mr_np, wt_np))
    ora = po.max_bound_batch(mr_np, wt_np)
    coll = pd.NativeCollective(engine)
    nb, roff, voff = engine.ragged_buffers(batch)
    cols = pg.Columns.allocate(515 * batch, 517 * batch, "cuda:0", 3, 5)  # the worst case
    res = torch.empty((batch,), dtype=torch.int64, device="cuda:0")
    cc, s = cols.as_c(), _lib.ShardC()
    st = lib.pg_max_bound_ragged_sharded_batch(coll._h, mr.data_ptr(), wt.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(),
                                               voff.data_ptr(), 3, 5, C.byref(cc), res.data_ptr(), C.byref(s), engine._stream())
    assert st == 0, lib.pg_last_error()
    torch.cuda.synchronize()
    assert (s.rank, s.world, s.gate_base, s.var_base, s.n_gates, s.n_vars) == (0, 1, 3, 5, ora["n_gates"], ora["n_vars"])
    got = cols.to_numpy()
    for name in COLS:
        n = ora["n_vars"] if name == "var_values" else ora["n_gates"]
        assert np.array_equal(got[name][:n], ora[name]), name
    assert np.array_equal(res.cpu().numpy().view(np.uint64), ora["result_vars"])
    c2, r2, info, gates, vars_ = pd.max_bound_ragged_sharded(engine, mr, wt, 3, 5, collective=coll)
    torch.cuda.synchronize()
    assert (gates, vars_, info.gate_base, info.var_base) == ([ora["n_gates"]], [ora["n_vars"]], 3, 5)
    got2 = c2.to_numpy()
    for name in COLS:
        assert np.array_equal(got2[name], ora[name]), name
    assert lib.pg_max_bound_ragged_sharded_plan(None, mr.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(), voff.data_ptr(), 3, 5,
                                                C.byref(s), None, None, engine._stream()) == 2
    # the gather pipeline's argument checks: nothing aborts, nothing is launched
    mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)
    h = C.c_void_p()
    assert lib.pg_range_check_gather_pipeline_create(coll._h, C.byref(mn.c), C.byref(mx.c), 0, 0, C.byref(h)) == 2  # chunk 0
    assert lib.pg_range_check_gather_pipeline_create(None, C.byref(mn.c), C.byref(mx.c), 8, 0, C.byref(h)) == 2
    assert lib.pg_range_check_gather_pipeline_create(coll._h, C.byref(mn.c), C.byref(mx.c), 8, 1, C.byref(h)) == 0, lib.pg_last_error()
    assert lib.pg_range_check_gather_pipeline_bytes_per_chunk(h) == 8 * engine.range_check_layout(mn, mx, 1).vars_per_item * 32
    nothing = _lib.CHUNK_CONSUMER(lambda *a: None)
    assert lib.pg_range_check_gather_pipeline_run(h, wt.data_ptr(), 12, 3, 5, nothing, None, engine._stream()) == 2  # not a multiple of the chunk
    assert lib.pg_range_check_gather_pipeline_run(h, None, 8, 3, 5, nothing, None, engine._stream()) == 2
    assert lib.pg_range_check_gather_pipeline_run(h, wt.data_ptr(), 0, 3, 5, nothing, None, engine._stream()) == 0  # an empty batch
    lib.pg_range_check_gather_pipeline_destroy(h)
    lib.pg_range_check_gather_pipeline_destroy(None)
    coll.close()


def _bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has 2+ GPUs")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--log2-batch", "10", "--no-cpu"])
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr and not r.stdout.strip(), (r.returncode, r.stdout, r.stderr[-2000:])


def test_bench_starts_the_ranks_it_was_asked_for():
    """`bench.py --gpus 2` started plainly: the parent starts two rank processes that run the real engine (sharing this
    box's one GPU: gloo rendezvous, the rehearsal backend) and rank 0 prints one line with n_gpus = 2 and twice the
    per-rank work"""
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "12", "--no-cpu"], {"PG_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["items_per_gpu"] == 4096
    rows = 2 * 2 * 4096 * 1031  # ranks x steps x witnesses x rows
    assert abs(line["value"] * line["ms_per_step"] * 2 / 1e3 - rows) < 1e-6 * rows
    assert "cpu_baseline" not in line and "secondary" not in line
