/* tests/cpp/fake_rccl.c -- TEST INFRASTRUCTURE, never shipped, never loaded unless PG_RCCL_LIB names it.
 *
 * A stand-in for the nine nccl* entry points libplonk_gadgets_hip binds at run time (csrc/capi_dist.inc), for boxes with ONE
 * GPU: the ranks are processes that share device 0, and an all-gather is a stream-ordered bounce through POSIX shared
 * memory.  It exists so that the world > 1 branches of the C-ABI multi-GPU slice (other ranks' parts of a chunk, the
 * regeneration of their rows, the totals exchange of ragged shards, bases of rank > 0) execute somewhere; it rehearses
 * ordering and arithmetic and measures nothing.
 *
 *   gcc -std=c11 -O1 -shared -fPIC -D_DEFAULT_SOURCE -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include tests/cpp/fake_rccl.c \
 *       -L /opt/rocm/lib -lamdhip64 -lrt -o tests/cpp/libfake_rccl.so
 *
 * ncclAllGather(send, recv, count, type, comm, stream), like the real one, returns at once and is ordered by `stream`:
 *     hipMemcpyAsync  send -> pinned staging                       (device to host, on `stream`)
 *     hipLaunchHostFunc: wait until the shared slot is free (every rank has taken what it held two exchanges ago), copy
 *         the staging buffer into this rank's part of the slot, publish "posted", wait until every rank has posted, copy
 *         all parts into the pinned receive staging, publish "taken"
 *     hipMemcpyAsync  staging -> recv + r * bytes, for every rank r (host to device, on `stream`)
 * in pieces of FAKE_RCCL_PIECE_BYTES (default 8 MiB).  Ranks issue their collectives in the same order (NCCL's rule too),
 * so exchange number s of one rank meets exchange number s of the others.  FAKE_RCCL_DELAY_US sleeps inside the host
 * function before the data is published: a consumer that fails to wait for the collective then reads stale bytes.
 * A wait that lasts longer than FAKE_RCCL_TIMEOUT_S (default 120) marks the communicator failed: calls return an error
 * instead of hanging the box.
 */
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FAKE_MAX_WORLD 8
#define NCCL_UNIQUE_ID_BYTES 128

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef int ncclDataType_t;  /* rccl.h: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8, bfloat16 9 */

struct shm_hdr {
    _Atomic uint32_t joined, left;
    _Atomic uint64_t posted[FAKE_MAX_WORLD]; /* number of the last exchange rank r has put into its slot */
    _Atomic uint64_t taken[FAKE_MAX_WORLD];  /* number of the last exchange rank r has copied out */
    _Atomic uint32_t failed;
    uint8_t pad[4096 - 2 * 4 - 2 * 8 * FAKE_MAX_WORLD - 4];
};

struct ncclComm {
    int rank, world;
    struct shm_hdr *shm;
    uint8_t *slots; /* [2][world][piece] */
    size_t shm_bytes, piece;
    uint8_t *stage; /* pinned: [1 + world][piece] */
    uint64_t seq;   /* exchanges enqueued so far */
    hipEvent_t last;
    int have_last;
    unsigned delay_us;
    double timeout_s;
};
typedef struct ncclComm *ncclComm_t;

static _Thread_local int g_group_depth;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static size_t env_size(const char *name, size_t dflt) {
    const char *v = getenv(name);
    return v && *v ? (size_t)strtoull(v, NULL, 10) : dflt;
}

static size_t type_bytes(ncclDataType_t t) {
    static const size_t sz[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};
    return t >= 0 && t < (int)(sizeof sz / sizeof sz[0]) ? sz[t] : 0;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
    case ncclSystemError: return "fake_rccl: shared memory / rendezvous failed or a peer timed out";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    default: return "fake_rccl: internal error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%ld_%lx%lx", (long)getpid(), (unsigned long)t.tv_sec, (unsigned long)t.tv_nsec);
    return ncclSuccess;
}

/* waits until every rank's counter has reached `want`; 0 = they have, -1 = timeout / a peer failed */
static int wait_all(struct ncclComm *c, _Atomic uint64_t *ctr, uint64_t want) {
    const double t0 = now_s();
    unsigned spins = 0;
    for (int r = 0; r < c->world; r++) {
        while (atomic_load_explicit(&ctr[r], memory_order_acquire) < want) {
            if (atomic_load_explicit(&c->shm->failed, memory_order_relaxed)) return -1;
            if ((++spins & 63) == 0) {
                if (now_s() - t0 > c->timeout_s) {
                    atomic_store(&c->shm->failed, 1);
                    fprintf(stderr, "fake_rccl: rank %d waited %.0f s for rank %d (exchange %llu)\n", c->rank, c->timeout_s, r,
                            (unsigned long long)want);
                    return -1;
                }
                usleep(50);
            }
        }
    }
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > FAKE_MAX_WORLD || rank < 0 || rank >= world) return ncclInvalidArgument;
    if (id.internal[0] != '/' || memchr(id.internal, 0, sizeof id.internal) == NULL) return ncclInvalidArgument;
    struct ncclComm *c = calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank;
    c->world = world;
    c->piece = env_size("FAKE_RCCL_PIECE_BYTES", 8u << 20);
    c->piece = (c->piece + 63) & ~(size_t)63;
    c->delay_us = (unsigned)env_size("FAKE_RCCL_DELAY_US", 0);
    c->timeout_s = (double)env_size("FAKE_RCCL_TIMEOUT_S", 120);
    c->shm_bytes = sizeof(struct shm_hdr) + 2 * (size_t)world * c->piece;
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) {
        perror("fake_rccl: shm_open / ftruncate");
        if (fd >= 0) close(fd);
        free(c);
        return ncclSystemError;
    }
    void *m = mmap(NULL, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        perror("fake_rccl: mmap");
        free(c);
        return ncclSystemError;
    }
    c->shm = m;
    c->slots = (uint8_t *)m + sizeof(struct shm_hdr);
    atomic_fetch_add(&c->shm->joined, 1);
    const double t0 = now_s();
    while (atomic_load(&c->shm->joined) < (uint32_t)world) {
        if (now_s() - t0 > c->timeout_s) {
            fprintf(stderr, "fake_rccl: rank %d: only %u of %d ranks joined\n", rank, atomic_load(&c->shm->joined), world);
            shm_unlink(id.internal);
            munmap(m, c->shm_bytes);
            free(c);
            return ncclSystemError;
        }
        usleep(200);
    }
    if (rank == 0) shm_unlink(id.internal); /* every rank has it mapped: nothing is left behind in /dev/shm whatever happens next */
    if (hipHostMalloc((void **)&c->stage, (size_t)(1 + world) * c->piece, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->last, hipEventDisableTiming) != hipSuccess) {
        munmap(m, c->shm_bytes);
        free(c);
        return ncclUnhandledCudaError;
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    if (c->have_last) (void)hipEventSynchronize(c->last);
    (void)hipEventDestroy(c->last);
    (void)hipHostFree(c->stage);
    atomic_fetch_add(&c->shm->left, 1);
    munmap(c->shm, c->shm_bytes);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *n) {
    if (!c || !n) return ncclInvalidArgument;
    *n = c->world;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) {
    if (!c || !r) return ncclInvalidArgument;
    *r = c->rank;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
    g_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (g_group_depth <= 0) return ncclInvalidArgument;
    g_group_depth--; /* the members of a group were enqueued as they came: every rank issues them in the same order */
    return ncclSuccess;
}

struct exchange {
    struct ncclComm *c;
    uint64_t seq;
    size_t bytes;
};

static void exchange_on_host(void *arg) {
    struct exchange *x = arg;
    struct ncclComm *c = x->c;
    const size_t P = c->piece;
    uint8_t *slot = c->slots + (size_t)(x->seq & 1) * (size_t)c->world * P;
    if (x->seq > 2 && wait_all(c, c->shm->taken, x->seq - 2) != 0) goto out; /* the slot still holds exchange seq - 2 */
    memcpy(slot + (size_t)c->rank * P, c->stage, x->bytes);
    if (c->delay_us) usleep(c->delay_us);
    atomic_store_explicit(&c->shm->posted[c->rank], x->seq, memory_order_release);
    if (wait_all(c, c->shm->posted, x->seq) != 0) goto out;
    for (int r = 0; r < c->world; r++) memcpy(c->stage + (size_t)(1 + r) * P, slot + (size_t)r * P, x->bytes);
    atomic_store_explicit(&c->shm->taken[c->rank], x->seq, memory_order_release);
out:
    free(x);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
    const size_t esz = type_bytes(type);
    if (!c || !esz || (count && (!send || !recv))) return ncclInvalidArgument;
    if (atomic_load(&c->shm->failed)) return ncclSystemError;
    const size_t bytes = count * esz;
    /* one staging buffer per communicator: a collective on another stream waits for the previous one */
    if (c->have_last && hipStreamWaitEvent(stream, c->last, 0) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t off = 0; off < bytes; off += c->piece) {
        const size_t n = bytes - off < c->piece ? bytes - off : c->piece;
        struct exchange *x = malloc(sizeof *x);
        if (!x) return ncclSystemError;
        x->c = c;
        x->seq = ++c->seq;
        x->bytes = n;
        if (hipMemcpyAsync(c->stage, (const uint8_t *)send + off, n, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipLaunchHostFunc(stream, exchange_on_host, x) != hipSuccess)
            return ncclUnhandledCudaError;
        for (int r = 0; r < c->world; r++)
            if (hipMemcpyAsync((uint8_t *)recv + (size_t)r * bytes + off, c->stage + (size_t)(1 + r) * c->piece, n, hipMemcpyHostToDevice,
                               stream) != hipSuccess)
                return ncclUnhandledCudaError;
    }
    if (hipEventRecord(c->last, stream) != hipSuccess) return ncclUnhandledCudaError;
    c->have_last = 1;
    return ncclSuccess;
}
