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
// the same from a scalar that lies in memory (a queued command's field), limb by limb: no copy of the aggregate
__device__ __forceinline__ void put_fr_at(uint4 *col, uint64_t i, const Fr *src) {
    const uint64_t l0 = src->l[0], l1 = src->l[1], l2 = src->l[2], l3 = src->l[3];
    col[2 * i] = make_uint4((uint32_t)l0, (uint32_t)(l0 >> 32), (uint32_t)l1, (uint32_t)(l1 >> 32));
    col[2 * i + 1] = make_uint4((uint32_t)l2, (uint32_t)(l2 >> 32), (uint32_t)l3, (uint32_t)(l3 >> 32));
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
// (one workgroup per launch: nothing is gained by leaving registers to a second one, and the compiler, aiming at the two
// workgroups per CU the LDS would allow, spilled)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void gate_queue_kernel(const GateCmd *cmds, uint32_t n, uint64_t first_var, uint32_t max_level,
                                                          const ComposerCols C) {
    __shared__ uint4 s_val[2 * kQueueRun];
    const uint32_t tid = threadIdx.x;
    // (a command is 224 bytes: what a phase needs of it is read when the phase needs it -- held in registers across the
    // level loop, the five selectors and the public input spilled)
    const GateCmd *mine = cmds + (tid < n ? tid : 0);
    uint32_t op = tid < n ? mine->op : (uint32_t)OP_ROW | OP_ROW_IN_PLACE;
    const bool in_place = (op & OP_ROW_IN_PLACE) != 0;
    op &= ~OP_ROW_IN_PLACE;
    const bool creates = tid < n && (op == OP_ADD || op == OP_MUL);
    const uint32_t level = creates ? mine->pad : 0;
    const uint64_t var = mine->var, wa = mine->a, wb = mine->b;
    if (tid < n) {
        if (op == OP_ADD_INPUT) {
            put_fr_at(C.vars, var, &mine->value);
            put_fr_at(s_val, var - first_var, &mine->value);
        } else if (!in_place) {
            const uint64_t gate = mine->gate;
            put_fr_at(C.q[0], gate, &mine->q_m);
            put_fr_at(C.q[1], gate, &mine->q_l);
            put_fr_at(C.q[2], gate, &mine->q_r);
            put_fr_at(C.q[3], gate, &mine->q_o);
            put_fr_at(C.q[4], gate, &mine->q_c);
            C.w[0][gate] = wa;
            C.w[1][gate] = wb;
            C.w[2][gate] = creates ? var : mine->c;
        }
    }
    __syncthreads();
    for (uint32_t lvl = 1; lvl <= max_level; lvl++) {
        if (level == lvl) {
            // an operand created by this run sits in LDS (its command has a lower level: done); any other is older
            const Fr a = wa >= first_var && wa - first_var < kQueueRun ? get_fr(s_val, wa - first_var) : get_fr(C.vars, wa);
            const Fr b = wb >= first_var && wb - first_var < kQueueRun ? get_fr(s_val, wb - first_var) : get_fr(C.vars, wb);
            Fr v = op == OP_ADD ? fr_add(fr_mul(mine->q_l, a), fr_mul(mine->q_r, b)) : fr_mul(fr_mul(mine->q_m, a), b);
            v = fr_add(fr_add(v, mine->q_c), mine->pi);
            put_fr(C.vars, var, v);
            put_fr(s_val, var - first_var, v);
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
__global__ __launch_bounds__(kThreads) void gate_batch_kernel(const GateBatch B_in, const ComposerCols C) {
    GateBatch B = B_in;
    const bool in_place = (B.op & OP_ROW_IN_PLACE) != 0;  // a witness refresh: the rows are in the columns already
    B.op &= ~OP_ROW_IN_PLACE;
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
        if (in_place) continue;
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

// largest entry of a device array of Variables -> out[0] (atomicMax; starts at 0): the batched appends check it against the
// composer's variable count before anything is appended (the reference panics on an unknown Variable).  In the same pass a
// 128-bit digest of the array -> out[1], out[2] (sums of two position-salted 64-bit mixes: order of summation does not matter,
// order of the entries does): an append whose rows depend on the array is signed with it (pg_composer_clear_witness).
__device__ __forceinline__ unsigned long long digest_mix(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(kThreads) void max_variable_kernel(const uint64_t *a, uint64_t n, unsigned long long *out,
                                                               unsigned long long salt) {
    unsigned long long m = 0, h1 = 0, h2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const unsigned long long v = a[i];
        m = v > m ? v : m;
        h1 += digest_mix(v + digest_mix(i + salt));
        h2 += digest_mix((v ^ 0x9e3779b97f4a7c15ull) * 0xff51afd7ed558ccdull + i * 0xc4ceb9fe1a85ec53ull + salt);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
        h1 += __shfl_xor(h1, d, 64);
        h2 += __shfl_xor(h2, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (m) atomicMax(out, m);
        atomicAdd(out + 1, h1);
        atomicAdd(out + 2, h2);
    }
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

// pg_composer_materialize in ONE pass: per row the seven constant columns (q_arith = 1, the rest 0, the value
// of the fourth wire = 0), w_4 = zero_var, and the three wire-VALUE columns out[i] = variables[w[i]] -- 328 B written and
// 24 B of indices + 96 B of assignments read per row.  Two lanes per row (16-byte halves, every wave store one contiguous
// KiB per column), two rows' worth per lane and pass: the six dependent loads (index, then assignment) of a pass are in
// flight together, and the constant stores keep the write queues busy meanwhile (as three kernels of one dependent chain per
// lane the gathers ran at 2.3 TB/s).  Any output pointer may be NULL.
struct MaterializeOut {
    uint4 *konst[7];  // q_4, q_arith, q_range, q_logic, q_fixed_group_add, q_variable_group_add, w_4_value
    uint4 *val[3];    // w_l_value, w_r_value, w_o_value
    uint64_t *w_4;
};
__global__ __launch_bounds__(kThreads) void materialize_kernel(const ComposerCols C, const MaterializeOut M, uint64_t row_begin,
                                                              uint64_t n_rows /* end of the range */, uint64_t zero_var) {
    constexpr uint64_t kPiece = 16384;  // 16-byte units per workgroup piece = 8192 rows
    constexpr int U = 2;            // (1 or 4, pieces of 64 Ki units: the same 5.9 TB/s of reads + writes, profiles/NOTES_r04.md)
    FrVec one;
    one.f = fr_one();
    const uint4 v1 = (threadIdx.x & 1) ? one.v[1] : one.v[0], v0 = make_uint4(0, 0, 0, 0);  // (a select: indexing the halves by the lane costs scratch)
    const uint64_t units = 2 * n_rows;
    for (uint64_t base = 2 * row_begin + (uint64_t)blockIdx.x * kPiece; base < units; base += (uint64_t)gridDim.x * kPiece) {
        const uint64_t end = base + kPiece < units ? base + kPiece : units;
        for (uint64_t i0 = base + threadIdx.x; i0 < end; i0 += (uint64_t)U * kThreads) {
            uint64_t idx[U][3];
            uint4 got[U][3];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = i0 + (uint64_t)u * kThreads, r = (i < end ? i : end - 1) >> 1;
#pragma unroll
                for (int k = 0; k < 3; k++) idx[u][k] = M.val[k] ? C.w[k][r] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = i0 + (uint64_t)u * kThreads;
#pragma unroll
                for (int k = 0; k < 3; k++) got[u][k] = C.vars[2 * idx[u][k] + (i & 1)];  // (a column not asked for: Variable 0, unused)
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = i0 + (uint64_t)u * kThreads;
                if (i < end) {
#pragma unroll
                    for (int k = 0; k < 7; k++)
                        if (M.konst[k]) store16(M.konst[k] + i, k == 1 ? v1 : v0);
                    if (M.w_4 && !(i & 1)) M.w_4[i >> 1] = zero_var;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t i = i0 + (uint64_t)u * kThreads;
                if (i < end) {
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        if (M.val[k]) store16(M.val[k] + i, got[u][k]);
                }
            }
        }
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
