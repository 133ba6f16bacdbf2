"""TEST INFRASTRUCTURE: a whole circuit on the host, the way oracle/composer.c would hold it after the same calls -- its four wire
columns and its dense assignment table -- assembled from the threaded gadget forms of oracle/fast.c, so that circuits of hundreds
of millions of rows fit the test's time.  The f-rows of that circuit (sigma, the materialised columns) then come from
oracle/fast.c's oracle_sigma_fast_* / oracle_materialize_fast.

Every method mirrors one append of the device composer (plonk_gadgets_amd.composer.StandardComposer) and the loop of reference
calls it stands for; tests/test_oracle_fast.py pins each of them to the same loop run call by call on oracle/composer.c +
oracle/gadgets.c (the faithful restatement of /root/reference/src/range.rs and src/scalar.rs)."""
import numpy as np

from oracle import pyoracle as po
from plonk_gadgets_amd import synth

WIRES = ("w_l", "w_r", "w_o")


def host_memory_available() -> int:
    """bytes of host memory this process may still take: /proc/meminfo's MemAvailable, and what is left of the cgroup's limit
    where there is one (a box that runs out kills the process -- and the box -- without a word)"""
    avail = 1 << 62
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) << 10
    except OSError:
        pass
    for limit, used in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                        ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            lim = open(limit).read().strip()
            if lim != "max":
                avail = min(avail, int(lim) - int(open(used).read()))
        except (OSError, ValueError):
            pass
    return avail


def host_bytes(cap_rows: int, cap_vars: int) -> int:
    """what a HostCircuit of this size and the oracle's sigma plan over it hold on the host: four wire columns, the assignments,
    first / last position and the carry of every Variable, staging buffers -- an upper estimate (16 GB of slack)"""
    return cap_rows * 32 + cap_vars * (32 + 40) + (16 << 30)


class HostCircuit:
    def __init__(self, cap_rows: int, cap_vars: int, threads: int = 1, dummy: bool = True):
        self.threads = threads
        self.w = [np.empty(cap_rows, dtype=np.uint64) for _ in range(4)]   # w_l, w_r, w_o, w_4
        self.values = np.empty((cap_vars, 4), dtype=np.uint64)
        ora = po.Composer(dummy=dummy)                                      # StandardComposer::new(): composer.c itself
        first, full = ora.export(), ora.full_columns()
        self.n, self.nv, self.zero_var = ora.n, ora.num_vars, int(ora.L.composer_zero_var(ora.c))
        for k, name in enumerate(WIRES):
            self.w[k][:self.n] = first[name]
        self.w[3][:self.n] = full["w_4"]
        self.values[:self.nv] = first["var_values"]
        self.q4 = {r: [int(x) for x in full["q_4"][r]] for r in range(self.n) if full["q_4"][r].any()}

    # -- views ------------------------------------------------------------------------------------------------------------
    def _out(self, rows: int, nvars: int):
        assert self.n + rows <= self.w[0].shape[0] and self.nv + nvars <= self.values.shape[0], "HostCircuit: capacity"
        out = {name: self.w[k][self.n:self.n + rows] for k, name in enumerate(WIRES)}
        out["var_values"] = self.values[self.nv:self.nv + nvars]
        self.w[3][self.n:self.n + rows] = self.zero_var                   # add / mul / poly_gate: the fourth wire is zero_var
        return out

    def wires(self):
        return [w[:self.n] for w in self.w]

    def table(self):
        return self.values[:self.nv]

    # -- appends ----------------------------------------------------------------------------------------------------------
    def add_input_batch(self, scalars: np.ndarray) -> int:
        """for s in scalars: add_input(s)"""
        k, first = scalars.shape[0], self.nv
        self.values[first:first + k] = scalars
        self.nv += k
        return first

    def gate_rows(self, a, b, c):
        """rows that create no Variable (constrain_to_constant, assert_equal, poly_gate, mul_gate, boolean_gate): wires only"""
        a, b, c = (np.asarray(x, dtype=np.uint64).reshape(-1) for x in (a, b, c))
        k = a.shape[0]
        out = self._out(k, 0)
        out["w_l"][:], out["w_r"][:], out["w_o"][:] = a, b, c
        self.n += k

    def range_check_batch(self, mn: int, mx: int, wit: np.ndarray, chunk: int = 1 << 12) -> np.ndarray:
        """for w in wit: allocate(w); range_check(mn, mx, w)"""
        batch, res = wit.shape[0], []
        n_bits = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit[:0], threads=1)["num_bits"]
        G, V = 4 * n_bits + 11, 2 * n_bits + 524
        for lo in range(0, batch, chunk):
            hi = min(batch, lo + chunk)
            out = self._out((hi - lo) * G, (hi - lo) * V)
            r = po.range_check_fast(synth.mont(mn), synth.mont(mx), np.ascontiguousarray(wit[lo:hi]), threads=self.threads,
                                    var_base=self.nv, out=out)
            res.append(r["result_vars"])
            self.n += (hi - lo) * G
            self.nv += (hi - lo) * V
        return np.concatenate(res) if res else np.zeros(0, np.uint64)

    def range_check_allocated_batch(self, mn: int, mx: int, wvars: np.ndarray, wit: np.ndarray, chunk: int = 1 << 12) -> np.ndarray:
        """for i: range_check(mn, mx, AllocatedScalar(wvars[i], wit[i])) -- the reference's own signature"""
        batch, res = wit.shape[0], []
        n_bits = po.range_check_fast(synth.mont(mn), synth.mont(mx), wit[:0], threads=1)["num_bits"]
        G, V = 4 * n_bits + 11, 2 * n_bits + 523
        for lo in range(0, batch, chunk):
            hi = min(batch, lo + chunk)
            out = self._out((hi - lo) * G, (hi - lo) * V)
            r = po.range_check_allocated_fast(synth.mont(mn), synth.mont(mx), np.ascontiguousarray(wit[lo:hi]),
                                              np.ascontiguousarray(wvars[lo:hi]), threads=self.threads, var_base=self.nv, out=out)
            res.append(r["result_vars"])
            self.n += (hi - lo) * G
            self.nv += (hi - lo) * V
        return np.concatenate(res) if res else np.zeros(0, np.uint64)

    def max_bound_ragged_batch(self, bounds: np.ndarray, wit: np.ndarray, chunk: int = 1 << 13):
        """for i: allocate(wit[i]); max_bound(bounds[i], .) -> (result Variables, ladder lengths)"""
        batch, res = wit.shape[0], []
        plan = po.max_bound_plan(bounds, threads=self.threads)
        nb, roff, voff = plan
        base = self.nv
        for lo in range(0, batch, chunk):
            hi = min(batch, lo + chunk)
            out = self._out(int(roff[hi] - roff[lo]), int(voff[hi] - voff[lo]))
            r = po.max_bound_fast(bounds, wit, plan, lo, hi, var_base=base, threads=self.threads, out=out)
            res.append(r["result_vars"])
            self.n += int(roff[hi] - roff[lo])
            self.nv += int(voff[hi] - voff[lo])
        return (np.concatenate(res) if res else np.zeros(0, np.uint64)), nb

    def scalar_mix_batch(self, v, y, s, a, b, chunk: int = 1 << 16):
        """for i: five add_input; is_non_zero; conditionally_select_one; maybe_equal -> (result Variables [batch, 2], error mask)"""
        batch, res = v.shape[0], []
        plan = po.scalar_mix_plan(v)
        roff, voff, err = plan
        base = self.nv
        for lo in range(0, batch, chunk):
            hi = min(batch, lo + chunk)
            out = self._out(int(roff[hi] - roff[lo]), int(voff[hi] - voff[lo]))
            r = po.scalar_mix_fast(v, y, s, a, b, plan, lo, hi, var_base=base, zero_var=self.zero_var, threads=self.threads, out=out)
            res.append(r["result_vars"])
            self.n += int(roff[hi] - roff[lo])
            self.nv += int(voff[hi] - voff[lo])
        return (np.concatenate(res) if res else np.zeros((0, 2), np.uint64)), err

    def range_check_loop_with_constrain(self, mn: int, mx: int, wit: np.ndarray) -> np.ndarray:
        """the reference tests' loop (/root/reference/tests/range_gadgets_tests.rs:29-44), once per witness:
        allocate(w); res = range_check(mn, mx, w); constrain_to_constant(res, outcome, None) -- the last one a row (res, res, res)
        and no Variable, so the items lie G + 1 rows apart"""
        batch = wit.shape[0]
        tmp = po.range_check_fast(synth.mont(mn), synth.mont(mx), np.ascontiguousarray(wit), threads=self.threads, var_base=self.nv)
        G, V = 4 * tmp["num_bits"] + 11, 2 * tmp["num_bits"] + 524
        out = self._out(batch * (G + 1), batch * V)
        for name in WIRES:
            rows = out[name].reshape(batch, G + 1)
            rows[:, :G] = tmp[name].reshape(batch, G)
            rows[:, G] = tmp["result_vars"]
        out["var_values"][:] = tmp["var_values"]
        self.n += batch * (G + 1)
        self.nv += batch * V
        return tmp["result_vars"]

    def small_batch(self, kind: str, a, b=None, c=None, selectors=None, chunk: int = 1 << 16):
        """for i: gadget(a[i], b[i]) on existing Variables -- kind: select_zero / select_one / maybe_equal / is_non_zero (-> error
        mask as well) / add / mul / rows (poly_gate and its special cases: selectors = q_m, q_l, q_r, q_o, q_c as ints)"""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        batch, res = a.shape[0], []
        sel = None if selectors is None else synth.scalars_from_ints(list(selectors))
        plan = po.is_non_zero_plan(a, self.table()) if kind == "is_non_zero" else None
        _, rows, nvars = po.SMALL_KINDS[kind]
        base, table = self.nv, self.table()
        for lo in range(0, batch, chunk):
            hi = min(batch, lo + chunk)
            G = int(plan[0][hi] - plan[0][lo]) if plan else rows * (hi - lo)
            V = int(plan[1][hi] - plan[1][lo]) if plan else nvars * (hi - lo)
            out = self._out(G, V)
            r = po.small_batch_fast(kind, a, b, c, table, lo, hi, var_base=base, zero_var=self.zero_var, selectors=sel, plan=plan,
                                    threads=self.threads, out=out)
            res.append(r["result_vars"])
            self.n += G
            self.nv += V
        res = np.concatenate(res) if res else np.zeros(0, np.uint64)
        return (res, plan[2]) if plan else res

    # -- the f-rows -------------------------------------------------------------------------------------------------------
    def sigma_plan(self, padded_n: int) -> po.SigmaFast:
        return po.SigmaFast(*self.wires(), padded_n=padded_n, n_vars=self.nv, threads=self.threads)

    def materialize(self, r0: int, r1: int, out=None) -> dict:
        return po.materialize_fast(*self.wires(), self.table(), r0, r1, q4=self.q4, threads=self.threads, out=out)
