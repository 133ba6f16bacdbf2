/* examples/c5_rank.c -- BASELINE config 5 from a compiled host: ONE PROCESS = ONE RANK = ONE GPU, plain C, no Python, no torch.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c5_rank.c \
 *       -L plonk_gadgets_amd -lplonk_gadgets_hip -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/plonk_gadgets_amd -Wl,-rpath,/opt/rocm/lib -o examples/c5_rank
 *
 *   c5_rank RANK WORLD ID_FILE TOTAL_PER_RANK CHUNK VARIABLES_ONLY [MAX_BITS]
 *
 * A batch of WORLD x TOTAL_PER_RANK witnesses of `allocate + range_check(0 <= w < 2^MAX_BITS)` is sharded as contiguous
 * witness ranges (rank r's item i is item r * TOTAL_PER_RANK + i) and streamed through the library's gather pipeline:
 * chunk k + 1 is emitted while chunk k is in ncclAllGather, and every rank ends with every rank's rows and variables
 * (VARIABLES_ONLY = 1: only the variable tables travel, the other ranks' rows are regenerated locally).  The communicator
 * id travels through ID_FILE (rank 0 writes it, the others wait for it).  Each rank folds everything it receives into a
 * digest, in (chunk, rank, array) order, and prints it: all ranks print the same digest, which the test compares with the
 * CPU oracle's for the same witnesses (tests/test_c_example.py).
 *
 * Witness g of the whole batch is the integer splitmix64(g + 1) mod 2^(MAX_BITS + 1): about half are in range.  MAX_BITS
 * defaults to 252 and may be at most 253 (a witness of 255 bits need not be below q).
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "plonk_gadgets_hip.h"

#define CK(x) do { pg_status s_ = (x); if (s_ != PG_OK) { fprintf(stderr, "rank %d: %s: %s (%s)\n", g_rank, #x, pg_status_string(s_), pg_last_error()); exit(1); } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_rank;

static uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

struct digest {
    uint64_t h;         /* FNV-1a over the 8-byte words received, in (chunk, rank, array) order */
    uint64_t words;
    uint64_t chunks;
    uint64_t *host;     /* staging buffer for one array */
    uint64_t host_words;
};

static void fold(struct digest *d, const void *dev, uint64_t words, void *stream) {
    if (words > d->host_words) {
        free(d->host);
        d->host = malloc(words * 8);
        d->host_words = words;
    }
    HK(hipMemcpyAsync(d->host, dev, words * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HK(hipStreamSynchronize((hipStream_t)stream));
    for (uint64_t i = 0; i < words; i++) {
        d->h ^= d->host[i];
        d->h *= 0x100000001b3ull;
    }
    d->words += words;
}

/* pg_chunk_consumer: parts[r] = rank r's chunk, complete for `stream` */
static void consume(void *user, uint64_t chunk_index, uint32_t world, const pg_columns *parts, uint64_t n_gates, uint64_t n_vars,
                    void *stream) {
    struct digest *d = user;
    (void)chunk_index;
    for (uint32_t r = 0; r < world; r++) {
        const pg_columns *p = &parts[r];
        fold(d, p->q_m, n_gates * 4, stream);
        fold(d, p->q_l, n_gates * 4, stream);
        fold(d, p->q_r, n_gates * 4, stream);
        fold(d, p->q_o, n_gates * 4, stream);
        fold(d, p->q_c, n_gates * 4, stream);
        fold(d, p->w_l, n_gates, stream);
        fold(d, p->w_r, n_gates, stream);
        fold(d, p->w_o, n_gates, stream);
        fold(d, p->var_values, n_vars * 4, stream);
    }
    d->chunks++;
}

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s RANK WORLD ID_FILE TOTAL_PER_RANK CHUNK VARIABLES_ONLY [MAX_BITS]\n", argv[0]);
        return 2;
    }
    const uint32_t rank = (uint32_t)atoi(argv[1]), world = (uint32_t)atoi(argv[2]);
    const char *id_file = argv[3];
    const uint64_t total = strtoull(argv[4], NULL, 10), chunk = strtoull(argv[5], NULL, 10);
    const uint32_t variables_only = (uint32_t)atoi(argv[6]);
    const unsigned max_bits = argc > 7 ? (unsigned)atoi(argv[7]) : 252;  /* (witnesses have MAX_BITS + 1 bits and must stay below q, a 255-bit number) */
    if (max_bits > 253) {
        fprintf(stderr, "MAX_BITS must be at most 253: witnesses of MAX_BITS + 1 bits have to be canonical (below q)\n");
        return 2;
    }
    g_rank = (int)rank;

    int ndev = 0;
    HK(hipGetDeviceCount(&ndev));
    const char *lr = getenv("LOCAL_RANK");
    const int device = (lr ? atoi(lr) : (int)rank) % (ndev > 0 ? ndev : 1);
    pg_engine *e = NULL;
    CK(pg_engine_create(device, &e));

    /* the communicator id: rank 0 makes it, the others read it when the file is complete (written under another name) */
    uint8_t id[PG_COMM_ID_BYTES];
    if (rank == 0) {
        CK(pg_comm_unique_id(id));
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, id_file) != 0) {
            perror("writing the communicator id");
            return 1;
        }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 6000 && !(f = fopen(id_file, "rb")); tries++) usleep(10000);
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) {
            fprintf(stderr, "rank %u: no communicator id in %s\n", rank, id_file);
            return 1;
        }
        fclose(f);
    }
    pg_comm *comm = NULL;
    CK(pg_comm_create(e, id, rank, world, &comm));

    /* public bounds and this rank's witnesses (canonical little-endian integers, decoded on the device) */
    pg_scalar min_range, max_range;
    pg_scalar_from_u64(0, &min_range);
    uint64_t raw[4] = {0, 0, 0, 0};
    raw[max_bits / 64] = 1ull << (max_bits % 64);
    pg_scalar_from_canonical(raw, &max_range);
    uint64_t *h_wit = calloc(total ? total : 1, 32);
    for (uint64_t i = 0; i < total; i++) {
        const uint64_t g = (uint64_t)rank * total + i;
        for (unsigned k = 0; k < 4; k++) {
            const unsigned lo = 64 * k;  /* bits [lo, lo + 64) of splitmix64-filled limbs, cut at MAX_BITS + 1 bits */
            uint64_t limb = splitmix64(4 * (g + 1) + k);
            if (max_bits + 1 <= lo) limb = 0;
            else if (max_bits + 1 < lo + 64) limb &= (1ull << (max_bits + 1 - lo)) - 1;
            h_wit[4 * i + k] = limb;
        }
    }
    void *d_raw = NULL;
    pg_scalar *d_wit = NULL;
    HK(hipMalloc(&d_raw, (total ? total : 1) * 32));
    HK(hipMalloc((void **)&d_wit, (total ? total : 1) * 32));
    HK(hipMemcpy(d_raw, h_wit, total * 32, hipMemcpyHostToDevice));
    uint64_t bad = 0;
    CK(pg_scalars_from_canonical_batch(e, d_raw, total, d_wit, NULL, &bad, NULL));
    if (bad) {
        fprintf(stderr, "rank %u: %llu witnesses are not canonical\n", rank, (unsigned long long)bad);
        return 1;
    }

    pg_gather_pipeline *pipe = NULL;
    CK(pg_range_check_gather_pipeline_create(comm, &min_range, &max_range, chunk, variables_only, &pipe));
    struct digest d = {0xcbf29ce484222325ull, 0, 0, NULL, 0};
    /* a fresh StandardComposer has 3 gates and 5 variables: the batch is numbered from there */
    CK(pg_range_check_gather_pipeline_run(pipe, d_wit, total, 3, 5, consume, &d, NULL));
    CK(pg_engine_sync(e, NULL));
    printf("rank %u of %u: %llu chunks, %llu words, %llu bytes per rank per chunk on the links, digest %016llx\n", rank, world,
           (unsigned long long)d.chunks, (unsigned long long)d.words,
           (unsigned long long)pg_range_check_gather_pipeline_bytes_per_chunk(pipe), (unsigned long long)d.h);

    pg_range_check_gather_pipeline_destroy(pipe);
    pg_comm_destroy(comm);
    pg_engine_destroy(e);
    free(d.host);
    free(h_wit);
    return 0;
}
