"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding arithmetic, the per-column all-gather and
the packed-chunk GatherPipeline.  The HIP engine cannot run here (no GPU), so a stand-in engine that answers with
the CPU oracle's rows takes its place -- what is under test is plonk_gadgets_amd.distributed, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plonk_gadgets_amd import synth

COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")
MN, MX = 50_000, 250_000


class OracleEngine:
    """test stand-in with Engine's range_check interface, rows from the oracle (relocated to the requested bases)"""
    device = torch.device("cpu")

    def range_check_layout(self, mn, mx, batch):
        import plonk_gadgets_amd as pg
        from plonk_gadgets_amd import _lib
        import ctypes as C
        lay = _lib.LayoutC()
        assert _lib.load().pg_range_check_layout(C.byref(mn.c), C.byref(mx.c), batch, C.byref(lay)) == 0
        return pg.Layout(lay.num_bits, lay.gates_per_item, lay.vars_per_item, lay.n_gates, lay.n_vars)

    def range_check_batch(self, mn, mx, witness, gate_base=0, var_base=0, out=None, result_vars=None):
        import plonk_gadgets_amd as pg
        from oracle import pyoracle as po
        ora = po.range_check_batch(mn.limbs(), mx.limbs(), witness.numpy().view(np.uint64))
        if out is None:
            out = pg.Columns.allocate(ora["n_gates"], ora["n_vars"], "cpu", gate_base, var_base)
        shift = np.uint64((var_base - 5) % 2**64)
        for k in COLS:
            a = ora[k] + shift if k.startswith("w_") else ora[k]
            getattr(out, k).copy_(torch.from_numpy(a.view(np.int64)))
        res = torch.from_numpy((ora["result_vars"] + shift).view(np.int64))
        if result_vars is not None:
            result_vars.copy_(res)
            res = result_vars
        return out, res


    def range_check_structure_batch(self, mn, mx, batch, gate_base, var_base, out):
        """rows of the oracle for ANY witnesses (zeros here): only the selector and wire columns are copied"""
        from oracle import pyoracle as po
        ora = po.range_check_batch(mn.limbs(), mx.limbs(), np.zeros((batch, 4), dtype=np.uint64))
        shift = np.uint64((var_base - 5) % 2**64)
        for k in COLS[:-1]:
            a = ora[k] + shift if k.startswith("w_") else ora[k]
            getattr(out, k).copy_(torch.from_numpy(a.view(np.int64)))
        return out

    # ---- ragged max_bound stand-ins (plan = per-item ladder bits and prefix sums, emit = oracle rows relocated) ----
    def ragged_buffers(self, batch):
        return (torch.empty((batch,), dtype=torch.int32), torch.empty((batch + 1,), dtype=torch.int64),
                torch.empty((batch + 1,), dtype=torch.int64))

    def max_bound_ragged_plan(self, max_range, nb, roff, voff):
        import plonk_gadgets_amd as pg
        ns = [pg.num_bits_closest_power_of_two(pg.BlsScalar.from_limbs(m) - pg.BlsScalar.one())
              for m in max_range.numpy().view(np.uint64)]
        nb.copy_(torch.tensor(ns, dtype=torch.int32))
        roff.copy_(torch.tensor(np.concatenate([[0], np.cumsum([2 * n + 5 for n in ns])]), dtype=torch.int64))
        voff.copy_(torch.tensor(np.concatenate([[0], np.cumsum([n + 262 for n in ns])]), dtype=torch.int64))
        return pg.Layout(0, 0, 0, int(roff[-1]), int(voff[-1]))

    def max_bound_ragged_emit(self, max_range, witness, nb, roff, voff, out, result_vars=None, gate_base=0, var_base=0):
        from oracle import pyoracle as po
        ora = po.max_bound_batch(max_range.numpy().view(np.uint64), witness.numpy().view(np.uint64))
        shift = np.uint64((var_base - 5) % 2**64)
        for k in COLS:
            a = ora[k] + shift if k.startswith("w_") else ora[k]
            getattr(out, k).copy_(torch.from_numpy(a.view(np.int64)))
        if result_vars is not None:
            result_vars.copy_(torch.from_numpy((ora["result_vars"] + shift).view(np.int64)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def witnesses(total):
    inside = synth.scalars_from_ints([MN + int(v) % (MX - MN) for v in synth.splitmix64(total // 2, 3)])
    return np.ascontiguousarray(np.concatenate([inside, synth.random_scalars(total - total // 2, 4)]))


def ragged_inputs():
    bounds = synth.scalars_from_ints([200, 2**128 - 1, 100, 2**200 + 7, 3])
    rwit = synth.scalars_from_ints([100, 2**127, 200, 5, 2])
    return bounds, rwit


def worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import plonk_gadgets_amd as pg
        from plonk_gadgets_amd import distributed as pd
        eng = OracleEngine()
        mn, mx = pg.BlsScalar.from_int(MN), pg.BlsScalar.from_int(MX)
        wit = witnesses(total)
        lo, hi = pd.shard_range(total, rank, world)
        local = torch.from_numpy(wit[lo:hi].view(np.int64))
        cols, res, info = pd.range_check_sharded(eng, mn, mx, local, total, gate_base=3, var_base=5)
        lay = eng.range_check_layout(mn, mx, 1)
        assert (info.lo, info.hi) == (lo, hi)
        assert info.gate_base == 3 + lo * lay.gates_per_item and info.var_base == 5 + lo * lay.vars_per_item
        ranges = [pd.shard_range(total, r, world) for r in range(world)]
        full, full_res = pd.gather_columns(cols, res, [(b - a) * lay.gates_per_item for a, b in ranges],
                                           [(b - a) * lay.vars_per_item for a, b in ranges])
        out = {k: getattr(full, k).numpy().view(np.uint64).copy() for k in COLS}
        out["result_vars"] = full_res.numpy().view(np.uint64).copy()
        # packed pipeline: equal shards of 4 items per rank, chunk 2 -> 2 chunks, one collective each
        per_rank, chunk = 4, 2
        wl = torch.from_numpy(wit[rank * per_rank:(rank + 1) * per_rank].view(np.int64))
        pipe = pd.GatherPipeline(eng, mn, mx, chunk)
        seen = []
        pipe.run(wl, per_rank, 3, 5, consume=lambda g, k: seen.append((k, g.clone())))
        out["pipe"] = [(k, g.numpy().view(np.uint64).copy()) for k, g in seen]
        out["pipe_lay"] = (pipe.lay.n_gates, pipe.lay.n_vars)
        # the same chunks with only the variable tables on the wire, the other rank's rows regenerated locally
        vpipe = pd.VariablesOnlyPipeline(eng, mn, mx, chunk)
        vseen = []
        vpipe.run(wl, per_rank, 3, 5, consume=lambda parts, k: vseen.append(
            (k, [{n: getattr(p, n).numpy().view(np.uint64).copy() for n in COLS} for p in parts])))
        out["vpipe"] = vseen
        out["vpipe_bytes"] = (vpipe.bytes_on_the_links_per_chunk(), pipe.bytes_per_chunk())
        # ragged max_bound: uneven shards (rank 0: 3 items, rank 1: 2), per-item bounds
        bounds, rwit = ragged_inputs()
        lo2, hi2 = (0, 3) if rank == 0 else (3, 5)
        rc, rr, info2, gates, vars_ = pd.max_bound_ragged_sharded(
            eng, torch.from_numpy(bounds[lo2:hi2].view(np.int64)), torch.from_numpy(rwit[lo2:hi2].view(np.int64)), 3, 5)
        rfull, rres = pd.gather_columns(rc, rr, gates, vars_)
        out["ragged"] = {k: getattr(rfull, k).numpy().view(np.uint64).copy() for k in COLS}
        out["ragged"]["result_vars"] = rres.numpy().view(np.uint64).copy()
        out["ragged_bases"] = (info2.gate_base, info2.var_base, gates, vars_)
        q.put((rank, out))
    except Exception as e:  # surface the failure instead of leaving the parent waiting on the queue
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 9])
def test_sharded_gather_matches_single_process(total):
    from oracle import pyoracle as po
    from plonk_gadgets_amd import distributed as pd
    world, port = 2, free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for r in got:
        assert "error" not in got[r], got[r]["error"]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    wit = witnesses(total)
    ora = po.range_check_batch(synth.mont(MN), synth.mont(MX), wit)
    for r in range(world):
        for k in COLS + ("result_vars",):
            assert np.array_equal(got[r][k], ora[k]), (r, k)
    # ragged sharded max_bound == the single-process oracle over all five items
    bounds, rwit = ragged_inputs()
    rora = po.max_bound_batch(bounds, rwit)
    for r in range(world):
        for k in COLS + ("result_vars",):
            assert np.array_equal(got[r]["ragged"][k], rora[k]), (r, k)
    ns = rora["num_bits"].tolist()
    first3_g, first3_v = sum(2 * n + 5 for n in ns[:3]), sum(n + 262 for n in ns[:3])
    assert got[0]["ragged_bases"][:2] == (3, 5) and got[1]["ragged_bases"][:2] == (3 + first3_g, 5 + first3_v)
    # packed chunks: gathered[src] of chunk k == the oracle rows of items src*4 + 2k .. +2, at global numbering
    G, V = 4 * ora["num_bits"] + 11, 2 * ora["num_bits"] + 524
    ng, nv = got[0]["pipe_lay"]
    off, sizes, words = pd.packed_layout(ng, nv)
    for r in range(world):
        assert [k for k, _ in got[r]["pipe"]] == [0, 1]
        for k, g in got[r]["pipe"]:
            assert g.shape == (world, words)
            for src in range(world):
                first = src * 4 + k * 2
                for name in COLS:
                    sec = g[src, off[name]:off[name] + sizes[name]]
                    if name == "var_values":
                        exp = ora[name][first * V:(first + 2) * V].reshape(-1)
                    elif name.startswith("w_"):
                        exp = ora[name][first * G:(first + 2) * G]
                    else:
                        exp = ora[name][first * G:(first + 2) * G].reshape(-1)
                    assert np.array_equal(sec, exp), (r, k, src, name)
        # variables-only pipeline: every part complete and equal to the oracle, from a fraction of the bytes
        assert [k for k, _ in got[r]["vpipe"]] == [0, 1]
        for k, parts in got[r]["vpipe"]:
            assert len(parts) == world
            for src in range(world):
                first = src * 4 + k * 2
                for name in COLS:
                    if name == "var_values":
                        exp = ora[name][first * V:(first + 2) * V]
                    else:
                        exp = ora[name][first * G:(first + 2) * G]
                    assert np.array_equal(parts[src][name], exp), (r, k, src, name)
        on_links, packed = got[r]["vpipe_bytes"]
        assert on_links == 2 * V * 32 and packed >= 2 * (G * 184 + V * 32)


def test_shard_range_partitions():
    from plonk_gadgets_amd import distributed as pd
    for total in (0, 1, 7, 8, 1 << 20, (1 << 23) + 5):
        for world in (1, 2, 3, 8):
            rs = [pd.shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def test_packed_layout_alignment():
    from plonk_gadgets_amd import distributed as pd
    for ng, nv in ((87, 562), (1031, 1034), (2, 3)):
        off, sizes, total = pd.packed_layout(ng, nv)
        assert all(o % 16 == 0 for o in off.values()) and total % 16 == 0  # sections start on 128-byte lines (int64 words)
        flat = torch.zeros(total, dtype=torch.int64)
        cols = pd.columns_in(flat, ng, nv)
        assert cols.q_m.shape == (ng, 4) and cols.w_o.shape == (ng,) and cols.var_values.shape == (nv, 4)
        assert cols.var_values.data_ptr() - flat.data_ptr() == off["var_values"] * 8


def test_shard_entry_points_of_the_c_abi():
    """pg_shard_range / pg_range_check_shard_layout / pg_columns_in_packed: host arithmetic, callable without a GPU"""
    import ctypes as C
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    lo, hi = C.c_uint64(), C.c_uint64()
    assert lib.pg_shard_range(10, 3, 3, C.byref(lo), C.byref(hi)) == 2  # rank >= world
    assert lib.pg_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) == 2
    assert lib.pg_shard_range(10, 2, 3, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (7, 10)
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
    s = _lib.ShardC()
    # BASELINE config 5: 2^23 witnesses over 8 ranks, rank 5
    assert lib.pg_range_check_shard_layout(C.byref(mn.c), C.byref(mx.c), 1 << 23, 5, 8, 3, 5, C.byref(s)) == 0
    assert (s.lo, s.hi) == (5 << 20, 6 << 20)
    assert (s.gate_base, s.var_base) == (3 + (5 << 20) * 1031, 5 + (5 << 20) * 1034)
    assert (s.n_gates, s.n_vars) == (1031 << 20, 1034 << 20)
    # the shards of all ranks tile the whole batch's numbering
    ends = []
    for r in range(3):
        assert lib.pg_range_check_shard_layout(C.byref(mn.c), C.byref(mx.c), 10, r, 3, 3, 5, C.byref(s)) == 0
        ends.append((s.gate_base, s.gate_base + s.n_gates, s.var_base, s.var_base + s.n_vars))
    assert ends[0][0] == 3 and ends[0][1] == ends[1][0] and ends[1][1] == ends[2][0] and ends[2][1] == 3 + 10 * 1031
    assert ends[0][2] == 5 and ends[0][3] == ends[1][2] and ends[1][3] == ends[2][2] and ends[2][3] == 5 + 10 * 1034
    # packed view: pointer arithmetic only
    buf = torch.zeros(4096, dtype=torch.int64)
    p, cc = _lib.PackedC(), _lib.ColumnsC()
    assert lib.pg_packed_layout(7, 9, C.byref(p)) == 0
    assert lib.pg_columns_in_packed(buf.data_ptr(), 7, 9, C.byref(cc)) == 0
    assert cc.q_m == buf.data_ptr() and cc.q_l - cc.q_m == 8 * 32 and cc.w_l - cc.q_c == 8 * 32  # 28 words, up to a line
    assert cc.w_r - cc.w_l == 8 * 16 and cc.var_values == buf.data_ptr() + 8 * p.var_words
    assert p.total_words == 5 * 32 + 3 * 16 + 48
    assert lib.pg_columns_in_packed(buf.data_ptr() + 8, 7, 9, C.byref(cc)) == 2  # misaligned


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`bench.py --gpus N` on a box with fewer than N GPUs fails loudly: never a silent 1-rank run"""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has 2+ GPUs")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "needs 2 GPUs" in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr)


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr)
