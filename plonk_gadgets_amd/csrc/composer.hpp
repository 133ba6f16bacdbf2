// composer.hpp -- device side of pg_composer, the device-resident counterpart of the slice of dusk-plonk's
// StandardComposer the reference's gadgets call (call sites: /root/reference/src/range.rs:42,
// 65,98,130,139,144,151 and src/scalar.rs:26,41-58,69-94,111-137; tests use constrain_to_constant and
// construct_dense_pi_vec).  The columns live in HBM and only ever grow; a single composer call is one tiny launch
// that appends one row and/or one variable, the gadgets append through the batched emitters (emit.hpp).
#pragma once

#include "emit.hpp"

namespace pg {

struct ComposerCols {
    uint4 *q[5];
    uint64_t *w[3];
    uint4 *vars;
};

enum GateOp : uint32_t { OP_ADD_INPUT = 0, OP_ROW = 1, OP_ADD = 2, OP_MUL = 3 };
// OR-ed into GateCmd::op: the row is already in the columns (a witness refresh, pg_composer_clear_witness) -- only the
// assignment the call creates is written
constexpr uint32_t OP_ROW_IN_PLACE = 0x80000000u;

// one composer call, passed by value as the kernel argument
struct GateCmd {
    uint32_t op;
    uint32_t pad;
    uint64_t gate;     // row to write (OP_ROW, OP_ADD, OP_MUL)
    uint64_t var;      // variable to create (OP_ADD_INPUT, OP_ADD, OP_MUL)
    uint64_t a, b, c;  // wires; for OP_ADD / OP_MUL the output wire is `var`
    Fr q_m, q_l, q_r, q_o, q_c;
    Fr pi;     // public input of the row (0 if none): enters the value of OP_ADD / OP_MUL outputs
    Fr value;  // OP_ADD_INPUT
};

__device__ __forceinline__ void put_fr(uint4 *col, uint64_t i, const Fr &f) {
    FrVec t;
    t.f = f;
    col[2 * i] = t.v[0];
    col[2 * i + 1] = t.v[1];
}
__device__ __forceinline__ Fr get_fr(const uint4 *col, uint64_t i) {
    FrVec t;
    t.v[0] = col[2 * i];
    t.v[1] = col[2 * i + 1];
    return t.f;
}

__global__ void gate_kernel(const GateCmd cmd, const ComposerCols C) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t op = cmd.op & ~OP_ROW_IN_PLACE;
    if (op == OP_ADD_INPUT) {
        put_fr(C.vars, cmd.var, cmd.value);
        return;
    }
    uint64_t out = cmd.c;
    if (op == OP_ADD || op == OP_MUL) {
        const Fr a = get_fr(C.vars, cmd.a), b = get_fr(C.vars, cmd.b);
        // big_add / big_mul with no fourth wire: c = q_l a + q_r b + q_c + pi   |   c = q_m a b + q_c + pi
        Fr v = op == OP_ADD ? fr_add(fr_mul(cmd.q_l, a), fr_mul(cmd.q_r, b)) : fr_mul(fr_mul(cmd.q_m, a), b);
        v = fr_add(fr_add(v, cmd.q_c), cmd.pi);
        put_fr(C.vars, cmd.var, v);
        out = cmd.var;
    }
    if (cmd.op & OP_ROW_IN_PLACE) return;
    put_fr(C.q[0], cmd.gate, cmd.q_m);
    put_fr(C.q[1], cmd.gate, cmd.q_l);
    put_fr(C.q[2], cmd.gate, cmd.q_r);
    put_fr(C.q[3], cmd.gate, cmd.q_o);
    put_fr(C.q[4], cmd.gate, cmd.q_c);
    C.w[0][cmd.gate] = cmd.a;
    C.w[1][cmd.gate] = cmd.b;
    C.w[2][cmd.gate] = out;
}

// A RUN of queued composer calls in ONE launch (pg_composer's command queue): cmds[i].pad = the command's level --
// 0 for calls that read no Variable of the run, else 1 + the highest level among the run's commands whose output it
// reads (computed on the host, where Variables are numbered).  One workgroup: rows and add_input values first, then the
// outputs of add / mul level by level; the assignments the run creates live in LDS (Variables first_var .. first_var +
// up to kQueueRun - 1) so that a command reads its predecessors' outputs without a trip through memory.
constexpr uint32_t kQueueRun = 1024;
__global__ __launch_bounds__(1024) void gate_queue_kernel(const GateCmd *cmds, uint32_t n, uint64_t first_var, uint32_t max_level,
                                                          const ComposerCols C) {
    __shared__ uint4 s_val[2 * kQueueRun];
    const uint32_t tid = threadIdx.x;
    GateCmd cmd{};
    if (tid < n) cmd = cmds[tid];
    const bool in_place = (cmd.op & OP_ROW_IN_PLACE) != 0;
    cmd.op &= ~OP_ROW_IN_PLACE;
    const bool creates = tid < n && (cmd.op == OP_ADD || cmd.op == OP_MUL);
    if (tid < n) {
        if (cmd.op == OP_ADD_INPUT) {
            put_fr(C.vars, cmd.var, cmd.value);
            put_fr(s_val, cmd.var - first_var, cmd.value);
        } else if (!in_place) {
            put_fr(C.q[0], cmd.gate, cmd.q_m);
            put_fr(C.q[1], cmd.gate, cmd.q_l);
            put_fr(C.q[2], cmd.gate, cmd.q_r);
            put_fr(C.q[3], cmd.gate, cmd.q_o);
            put_fr(C.q[4], cmd.gate, cmd.q_c);
            C.w[0][cmd.gate] = cmd.a;
            C.w[1][cmd.gate] = cmd.b;
            C.w[2][cmd.gate] = creates ? cmd.var : cmd.c;
        }
    }
    __syncthreads();
    for (uint32_t lvl = 1; lvl <= max_level; lvl++) {
        if (creates && cmd.pad == lvl) {
            // an operand created by this run sits in LDS (its command has a lower level: done); any other is older
            const Fr a = cmd.a >= first_var && cmd.a - first_var < kQueueRun ? get_fr(s_val, cmd.a - first_var) : get_fr(C.vars, cmd.a);
            const Fr b = cmd.b >= first_var && cmd.b - first_var < kQueueRun ? get_fr(s_val, cmd.b - first_var) : get_fr(C.vars, cmd.b);
            Fr v = cmd.op == OP_ADD ? fr_add(fr_mul(cmd.q_l, a), fr_mul(cmd.q_r, b)) : fr_mul(fr_mul(cmd.q_m, a), b);
            v = fr_add(fr_add(v, cmd.q_c), cmd.pi);
            put_fr(C.vars, cmd.var, v);
            put_fr(s_val, cmd.var - first_var, v);
        }
        __syncthreads();
    }
}

// the same calls over device arrays of Variables with one set of selectors: item i writes row gate0 + i; OP_ADD / OP_MUL
// also create Variable var0 + i = q_l a + q_r b + q_c  /  q_m a b + q_c (read from the composer's own table) on the
// output wire.  c may be NULL for OP_ADD / OP_MUL.
struct GateBatch {
    uint32_t op;
    uint32_t pad;
    uint64_t gate0, var0, batch;
    const uint64_t *a, *b, *c;
    uint64_t *out_vars;  // OP_ADD / OP_MUL: the new Variables (may be NULL)
    Fr q_m, q_l, q_r, q_o, q_c;
};
__global__ __launch_bounds__(kThreads) void gate_batch_kernel(const GateBatch B, const ComposerCols C) {
    FrVec q[5];
    q[0].f = B.q_m; q[1].f = B.q_l; q[2].f = B.q_r; q[3].f = B.q_o; q[4].f = B.q_c;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < B.batch; i += (uint64_t)gridDim.x * kThreads) {
        const uint64_t a = B.a[i], b = B.b[i], g = B.gate0 + i;
        uint64_t out = B.c ? B.c[i] : 0;
        if (B.op == OP_ADD || B.op == OP_MUL) {
            const Fr va = get_fr(C.vars, a), vb = get_fr(C.vars, b);
            const Fr v = B.op == OP_ADD ? fr_add(fr_add(fr_mul(B.q_l, va), fr_mul(B.q_r, vb)), B.q_c)
                                        : fr_add(fr_mul(fr_mul(B.q_m, va), vb), B.q_c);
            out = B.var0 + i;
            put_fr(C.vars, out, v);
            if (B.out_vars) B.out_vars[i] = out;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            C.q[k][2 * g] = q[k].v[0];
            C.q[k][2 * g + 1] = q[k].v[1];
        }
        C.w[0][g] = a;
        C.w[1][g] = b;
        C.w[2][g] = out;
    }
}

// largest entry of a device array of Variables -> *out (atomicMax; *out starts at 0): the batched appends check it
// against the composer's variable count before anything is appended (the reference panics on an unknown Variable)
__global__ __launch_bounds__(kThreads) void max_variable_kernel(const uint64_t *a, uint64_t n, unsigned long long *out) {
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) m = a[i] > m ? a[i] : m;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// small host blob -> device staging buffer (inputs of single-gadget calls: scalars, Variables, offsets)
struct StageBlob {
    uint64_t w[40];
};
__global__ void stage_kernel(const StageBlob blob, uint64_t *dst, uint32_t n) {
    if (threadIdx.x < n) dst[threadIdx.x] = blob.w[threadIdx.x];
}

// rows whose fourth wire is live (q_4 != 0 or w_4 != zero_var): only the initial dummy constraints on this path
struct FourthWire {
    uint64_t gate, w_4;
    Fr q_4;
};

// q_m a b + q_l a + q_r b + q_o c + q_4 d + q_c + PI == 0 on every row; first unsatisfied row -> *first_bad
// var_base: Variable index of C.vars[0] (0 for a composer; the call's var_base for the columns of one batch call);
// a wire outside [var_base, var_base + n_vars) marks the row as bad instead of being dereferenced
// (a wire equal to zero_var reads as 0 even when zero_var lies outside the table: assert_equal's output wire)
__global__ __launch_bounds__(kThreads) void check_kernel(const ComposerCols C, uint64_t n, uint64_t var_base, uint64_t n_vars,
                                                        uint64_t zero_var, const uint64_t *pi_gate, const uint4 *pi_val,
                                                        uint32_t n_pi, const FourthWire *fw, uint32_t n_fw,
                                                        unsigned long long *first_bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        Fr val[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint64_t w = C.w[k][i], rel = w - var_base;
            if (rel < n_vars) val[k] = get_fr(C.vars, rel);
            else if (w == zero_var) val[k] = fr_zero();
            else ok = false;
        }
        if (!ok) {
            atomicMin(first_bad, (unsigned long long)i);
            continue;
        }
        const Fr a = val[0], b = val[1], c = val[2];
        Fr t = fr_mul(fr_mul(get_fr(C.q[0], i), a), b);
        t = fr_add(t, fr_mul(get_fr(C.q[1], i), a));
        t = fr_add(t, fr_mul(get_fr(C.q[2], i), b));
        t = fr_add(t, fr_mul(get_fr(C.q[3], i), c));
        t = fr_add(t, get_fr(C.q[4], i));
        // sparse public inputs, sorted by gate (rows are appended in order): binary search
        uint32_t lo = 0, hi = n_pi;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pi_gate[mid] < i) lo = mid + 1; else hi = mid;
        }
        if (lo < n_pi && pi_gate[lo] == i) t = fr_add(t, get_fr(pi_val, lo));
        for (uint32_t k = 0; k < n_fw; k++)
            if (fw[k].gate == i) t = fr_add(t, fr_mul(fw[k].q_4, get_fr(C.vars, fw[k].w_4)));
        if (!fr_is_zero(t)) atomicMin(first_bad, (unsigned long long)i);
    }
}

// ---- the rest of a prover-ready row ------------------------------------
// constant columns (the selectors this path never switches on, q_arith = 1, the value column of a fourth wire that is
// zero_var on every row): up to 7 columns per launch, each workgroup writing contiguous 1 MiB pieces of each
struct ConstCols {
    uint4 *p[7];
    uint32_t n, one_mask;  // column k holds 1 (Montgomery R) if bit k of one_mask is set, else 0
};
__global__ __launch_bounds__(kThreads) void fill_columns_kernel(const ConstCols K, uint64_t n_rows) {
    constexpr uint64_t kPiece = 65536;
    FrVec one;
    one.f = fr_one();
    const uint4 v1 = one.v[threadIdx.x & 1], v0 = make_uint4(0, 0, 0, 0);
    for (uint64_t base = (uint64_t)blockIdx.x * kPiece; base < 2 * n_rows; base += (uint64_t)gridDim.x * kPiece) {
        const uint64_t end = base + kPiece < 2 * n_rows ? base + kPiece : 2 * n_rows;
        for (uint32_t k = 0; k < K.n; k++) {
            const uint4 v = K.one_mask >> k & 1 ? v1 : v0;
            for (uint64_t i = base + threadIdx.x; i < end; i += kThreads) store16(K.p[k] + i, v);
        }
    }
}
__global__ __launch_bounds__(kThreads) void fill_u64_kernel(uint64_t *dst, uint64_t n, uint64_t value) {
    constexpr uint64_t kPiece = 131072;
    for (uint64_t base = (uint64_t)blockIdx.x * kPiece; base < n; base += (uint64_t)gridDim.x * kPiece) {
        const uint64_t end = base + kPiece < n ? base + kPiece : n;
        for (uint64_t i = base + threadIdx.x; i < end; i += kThreads) dst[i] = value;
    }
}
// rows with a live fourth wire: q_4, w_4 and the value of w_4 (each may be NULL)
__global__ void patch_fourth_kernel(uint4 *q_4, uint64_t *w_4, uint4 *w_4_value, const uint4 *vars, const FourthWire *fw,
                                    uint32_t n_fw) {
    if (threadIdx.x < n_fw) {
        const FourthWire f = fw[threadIdx.x];
        if (q_4) put_fr(q_4, f.gate, f.q_4);
        if (w_4) w_4[f.gate] = f.w_4;
        if (w_4_value) put_fr(w_4_value, f.gate, get_fr(vars, f.w_4));
    }
}
// wire VALUE column: out[i] = variables[w[i]] (what the prover interpolates); 16 B per lane, contiguous pieces
__global__ __launch_bounds__(kThreads) void gather_wire_values_kernel(const uint64_t *w, const uint4 *vars, uint64_t n_rows,
                                                                     uint4 *out) {
    constexpr uint64_t kPiece = 32768;
    for (uint64_t base = (uint64_t)blockIdx.x * kPiece; base < 2 * n_rows; base += (uint64_t)gridDim.x * kPiece) {
        const uint64_t end = base + kPiece < 2 * n_rows ? base + kPiece : 2 * n_rows;
        for (uint64_t i = base + threadIdx.x; i < end; i += kThreads) store16(out + i, vars[2 * w[i >> 1] + (i & 1)]);
    }
}
// dense public-input vector from the sparse store (construct_dense_pi_vec)
__global__ void scatter_pi_kernel(uint4 *dense, const uint64_t *pi_gate, const uint4 *pi_val, uint32_t n_pi) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_pi) {
        dense[2 * pi_gate[k]] = pi_val[2 * k];
        dense[2 * pi_gate[k] + 1] = pi_val[2 * k + 1];
    }
}

}  // namespace pg
