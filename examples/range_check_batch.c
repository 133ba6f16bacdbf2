/* examples/range_check_batch.c -- the C ABI from plain C: no Python, no torch, no C++.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/range_check_batch.c \
 *       -L plonk_gadgets_amd -lplonk_gadgets_hip -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/plonk_gadgets_amd -Wl,-rpath,/opt/rocm/lib -o examples/range_check_batch
 *
 * Emits  for w in witnesses { allocate(w); range_check(50 000 <= w < 250 000) }  for the eight witnesses of the
 * reference's own test (tests/range_gadgets_tests.rs:120-169, the 2^126 case replaced by an in-range value), checks
 * on the device that every emitted row is satisfied, and prints the 0/1 outcome of each item.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "plonk_gadgets_hip.h"

#define CK(x) do { pg_status s_ = (x); if (s_ != PG_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, pg_status_string(s_), pg_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(void) {
    const uint64_t values[8] = {50001, 250001, 250000, 249000, 50000, 49999, 123456, 18598};
    const int expected[8] = {1, 0, 0, 1, 1, 0, 1, 0};
    enum { BATCH = 8 };
    pg_engine *e = NULL;
    CK(pg_engine_create(0, &e));

    pg_scalar min_range, max_range, witness[BATCH];
    pg_scalar_from_u64(50000, &min_range);
    pg_scalar_from_u64(250000, &max_range);
    for (int i = 0; i < BATCH; i++) pg_scalar_from_u64(values[i], &witness[i]);

    pg_layout lay;
    CK(pg_range_check_layout(&min_range, &max_range, BATCH, &lay));
    printf("ladder bits %llu, %llu rows + %llu variables per item\n", (unsigned long long)lay.num_bits,
           (unsigned long long)lay.gates_per_item, (unsigned long long)lay.vars_per_item);

    pg_columns cols;
    pg_scalar *d_witness;
    pg_variable *d_result;
    /* the nine arrays in ONE block, where the library says they should lie relative to each other (a stride of 0: back to
     * back on 2-MiB boundaries; a circuit of GBs on an MI355X would ask for 24 GiB between the selector columns) */
    uint64_t off[9], block_bytes;
    unsigned char *block;
    CK(pg_columns_slab_layout(lay.n_gates, lay.n_vars, 0, off, &block_bytes));
    HK(hipMalloc((void **)&block, block_bytes));
    cols.q_m = (pg_scalar *)(block + off[0]); cols.q_l = (pg_scalar *)(block + off[1]); cols.q_r = (pg_scalar *)(block + off[2]);
    cols.q_o = (pg_scalar *)(block + off[3]); cols.q_c = (pg_scalar *)(block + off[4]);
    cols.w_l = (pg_variable *)(block + off[5]); cols.w_r = (pg_variable *)(block + off[6]); cols.w_o = (pg_variable *)(block + off[7]);
    cols.var_values = (pg_scalar *)(block + off[8]);
    HK(hipMalloc((void **)&d_witness, sizeof witness));
    HK(hipMalloc((void **)&d_result, BATCH * sizeof(pg_variable)));
    HK(hipMemcpy(d_witness, witness, sizeof witness, hipMemcpyHostToDevice));

    /* a fresh StandardComposer has 3 gates and 5 variables: the batch is numbered from there */
    const uint64_t gate_base = 3, var_base = 5;
    CK(pg_range_check_batch(e, &min_range, &max_range, d_witness, BATCH, gate_base, var_base, &cols, d_result, NULL));
    int64_t bad = 0;
    CK(pg_check_rows(e, &cols, lay.n_gates, var_base, lay.n_vars, 0, &bad, NULL));
    printf("first unsatisfied row: %lld\n", (long long)bad);

    pg_variable result[BATCH];
    HK(hipMemcpy(result, d_result, sizeof result, hipMemcpyDeviceToHost));
    int ok = bad == -1;
    for (int i = 0; i < BATCH; i++) {
        pg_scalar v;
        uint64_t raw[4];
        HK(hipMemcpy(&v, cols.var_values + (result[i] - var_base), 32, hipMemcpyDeviceToHost));
        pg_scalar_to_canonical(&v, raw);
        printf("range_check(%llu) -> Variable(%llu) = %llu\n", (unsigned long long)values[i], (unsigned long long)result[i],
               (unsigned long long)raw[0]);
        ok = ok && raw[0] == (uint64_t)expected[i] && !raw[1] && !raw[2] && !raw[3];
    }
    pg_engine_destroy(e);
    puts(ok ? "OK" : "MISMATCH");
    return ok ? 0 : 2;
}
