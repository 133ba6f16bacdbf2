// capi.hip -- the C ABI of libplonk_gadgets_hip.so (include/plonk_gadgets_hip.h).
// Host-side validation and layout arithmetic + kernel launches.  There is no
// CPU fallback anywhere in this file: a call either runs on the GPU or
// returns an error status.
#include "../../include/plonk_gadgets_hip.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "fr.hpp"
#include "range_check.hpp"

namespace {

thread_local std::string g_last_error;

pg_status fail(pg_status s, const std::string &msg) {
    g_last_error = msg;
    return s;
}

#define PG_HIP_TRY(expr)                                                                          \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(PG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

inline pg::Fr to_fr(const pg_scalar *s) {
    pg::Fr f;
    std::memcpy(f.l, s->l, sizeof f.l);
    return f;
}
inline void from_fr(const pg::Fr &f, pg_scalar *out) { std::memcpy(out->l, f.l, sizeof f.l); }

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// a BlsScalar is always fully reduced; reject limbs >= q instead of computing garbage
bool is_reduced(const pg::Fr &f) {
    const uint64_t Q[4] = {PG_Q0, PG_Q1, PG_Q2, PG_Q3};
    for (int i = 3; i >= 0; i--) {
        if (f.l[i] < Q[i]) return true;
        if (f.l[i] > Q[i]) return false;
    }
    return false;
}

pg_status check_columns(const pg_columns *c) {
    if (!c) return fail(PG_ERR_INVALID_ARGUMENT, "columns is NULL");
    const void *sc[6] = {c->q_m, c->q_l, c->q_r, c->q_o, c->q_c, c->var_values};
    for (const void *p : sc)
        if (!p || !aligned(p, 16)) return fail(PG_ERR_INVALID_ARGUMENT, "scalar column NULL or not 16-byte aligned");
    const void *wc[3] = {c->w_l, c->w_r, c->w_o};
    for (const void *p : wc)
        if (!p || !aligned(p, 8)) return fail(PG_ERR_INVALID_ARGUMENT, "wire column NULL or not 8-byte aligned");
    return PG_OK;
}

}  // namespace

struct pg_engine {
    int device = -1;
    int num_cus = 0;
    uint4 *d_pow2 = nullptr;  // mont(2^i), i < 256
};

extern "C" {

const char *pg_status_string(pg_status s) {
    switch (s) {
        case PG_OK: return "ok";
        case PG_ERR_NON_EXISTING_INVERSE: return "non-existing inverse";
        case PG_ERR_INVALID_ARGUMENT: return "invalid argument";
        case PG_ERR_NO_DEVICE: return "no usable gfx950 device";
        case PG_ERR_HIP: return "HIP runtime error";
        case PG_ERR_CAPACITY: return "composer capacity exceeded";
    }
    return "unknown status";
}

const char *pg_last_error(void) { return g_last_error.c_str(); }
const char *pg_build_arch(void) { return "gfx950"; }

pg_status pg_engine_create(int device, pg_engine **out) {
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PG_ERR_NO_DEVICE, "hipGetDeviceCount found no device: this library has no CPU path");
    if (device < 0 || device >= count) return fail(PG_ERR_INVALID_ARGUMENT, "device index out of range");
    hipDeviceProp_t prop;
    PG_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PG_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", library is built for gfx950 only");
    PG_HIP_TRY(hipSetDevice(device));
    pg_engine *e = new (std::nothrow) pg_engine();
    if (!e) return fail(PG_ERR_HIP, "out of host memory");
    e->device = device;
    e->num_cus = prop.multiProcessorCount;
    if (hipMalloc(&e->d_pow2, 256 * 2 * sizeof(uint4)) != hipSuccess) {
        delete e;
        return fail(PG_ERR_HIP, "hipMalloc(pow2 table) failed");
    }
    hipLaunchKernelGGL(pg::pow2_table_kernel, dim3(1), dim3(64), 0, nullptr, e->d_pow2);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) {
        hipFree(e->d_pow2);
        delete e;
        return fail(PG_ERR_HIP, std::string("pow2 table kernel: ") + hipGetErrorString(err));
    }
    *out = e;
    return PG_OK;
}

void pg_engine_destroy(pg_engine *e) {
    if (!e) return;
    hipSetDevice(e->device);
    hipFree(e->d_pow2);
    delete e;
}

pg_status pg_engine_sync(pg_engine *e, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    PG_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PG_OK;
}

/* ---- host scalar helpers ------------------------------------------------ */
void pg_scalar_from_u64(uint64_t v, pg_scalar *out) { from_fr(pg::fr_from_u64(v), out); }
void pg_scalar_from_canonical(const uint64_t raw[4], pg_scalar *out) {
    from_fr(pg::fr_to_mont(pg::Fr{{raw[0], raw[1], raw[2], raw[3]}}), out);
}
void pg_scalar_to_canonical(const pg_scalar *s, uint64_t raw[4]) {
    pg::Fr c = pg::fr_from_mont(to_fr(s));
    std::memcpy(raw, c.l, sizeof c.l);
}
void pg_scalar_add(const pg_scalar *a, const pg_scalar *b, pg_scalar *out) { from_fr(pg::fr_add(to_fr(a), to_fr(b)), out); }
void pg_scalar_sub(const pg_scalar *a, const pg_scalar *b, pg_scalar *out) { from_fr(pg::fr_sub(to_fr(a), to_fr(b)), out); }
void pg_scalar_neg(const pg_scalar *a, pg_scalar *out) { from_fr(pg::fr_neg(to_fr(a)), out); }
void pg_scalar_mul(const pg_scalar *a, const pg_scalar *b, pg_scalar *out) { from_fr(pg::fr_mul(to_fr(a), to_fr(b)), out); }
uint64_t pg_bits_count(const pg_scalar *s) { return pg::bits_count(to_fr(s)); }
uint64_t pg_num_bits_closest_power_of_two(const pg_scalar *s) { return pg::num_bits_closest_power_of_two(to_fr(s)); }

/* ---- range_check ---------------------------------------------------------- */
pg_status pg_range_check_layout(const pg_scalar *min_range, const pg_scalar *max_range, uint64_t batch, pg_layout *out) {
    if (!min_range || !max_range || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    pg::Fr mn = to_fr(min_range), mx = to_fr(max_range);
    if (!is_reduced(mn) || !is_reduced(mx)) return fail(PG_ERR_INVALID_ARGUMENT, "bound is not a reduced BlsScalar");
    // range.rs:87-90
    uint64_t n = pg::num_bits_closest_power_of_two(pg::fr_sub(mx, pg::fr_one()));
    out->num_bits = n;
    out->gates_per_item = 4 * n + 11;
    out->vars_per_item = 2 * n + 524;  // 2n+523 of range_check + 1 of allocate
    out->n_gates = out->gates_per_item * batch;
    out->n_vars = out->vars_per_item * batch;
    return PG_OK;
}

pg_status pg_range_check_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                               const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                               const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    pg_layout lay;
    pg_status st = pg_range_check_layout(min_range, max_range, batch, &lay);
    if (st != PG_OK) return st;
    if (batch == 0) return PG_OK;
    if (!d_witness || !aligned(d_witness, 16)) return fail(PG_ERR_INVALID_ARGUMENT, "d_witness NULL or not 16-byte aligned");
    if (d_result_vars && !aligned(d_result_vars, 8)) return fail(PG_ERR_INVALID_ARGUMENT, "d_result_vars misaligned");
    if ((st = check_columns(out)) != PG_OK) return st;
    if (lay.num_bits < 2 || lay.num_bits > 255) return fail(PG_ERR_INVALID_ARGUMENT, "ladder length out of range");

    constexpr int W = 16;
    const uint64_t tiles = (batch + W - 1) / W;
    if (tiles > 0xffffffffull || lay.n_gates > (1ull << 40)) return fail(PG_ERR_INVALID_ARGUMENT, "batch too large for one call");

    pg::RangeCheckArgs A;
    A.min_range = to_fr(min_range);
    A.max_range = to_fr(max_range);
    A.n = (uint32_t)lay.num_bits;
    A.tiles = (uint32_t)tiles;
    A.batch = batch;
    A.gate_base = gate_base;
    A.var_base = var_base;
    A.q[0] = reinterpret_cast<uint4 *>(out->q_m);
    A.q[1] = reinterpret_cast<uint4 *>(out->q_l);
    A.q[2] = reinterpret_cast<uint4 *>(out->q_r);
    A.q[3] = reinterpret_cast<uint4 *>(out->q_o);
    A.q[4] = reinterpret_cast<uint4 *>(out->q_c);
    A.w[0] = out->w_l;
    A.w[1] = out->w_r;
    A.w[2] = out->w_o;
    A.vars = reinterpret_cast<uint4 *>(out->var_values);
    A.witness = reinterpret_cast<const uint4 *>(d_witness);
    A.result_vars = d_result_vars;
    A.pow2 = e->d_pow2;

    PG_HIP_TRY(hipSetDevice(e->device));
    const uint32_t max_blocks = (uint32_t)e->num_cus * 8;
    const uint32_t grid = tiles < max_blocks ? (uint32_t)tiles : max_blocks;
    hipLaunchKernelGGL(pg::range_check_kernel<W>, dim3(grid), dim3(pg::kThreads), 0, static_cast<hipStream_t>(stream), A);
    PG_HIP_TRY(hipGetLastError());
    return PG_OK;
}

}  // extern "C"
