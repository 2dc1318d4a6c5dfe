/*
 * mock_rccl.c -- a stand-in for librccl on a ONE-GPU test box (test infrastructure only).
 *
 * RCCL refuses two ranks on one device, and the GPU box of the test pool has one device, so the in-library RCCL
 * path of libtsdf_hip.so (csrc/rccl_dyn.cpp binds librccl at run time; TSDF_RCCL_LIBRARY names the library) can
 * only be driven with one rank by the real library.  This file implements the five entry points the product binds
 * (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclCommDestroy, ncclGetErrorString), with rccl.h's ABI, over a
 * POSIX shared-memory segment: the all-reduce synchronises the stream, copies the device buffer to the host, adds
 * the ranks' rows in rank order and copies the sum back.  What it exercises is the PRODUCT's code around the
 * collective with N > 1 ranks (communicator set-up from a broadcast id, the tracker's device-side result row feeding
 * the collective, the publish kernel, host polling, sharded trajectory = single-rank trajectory); it says nothing
 * about RCCL itself or about xGMI.
 *
 * Build:  make mock_rccl   (gcc, links libamdhip64 for hipMemcpy / hipStreamSynchronize)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
enum { kMaxCount = 64, kSlotBytes = 1024 };

typedef struct {
    int nranks, rank;
    char* base;
    size_t bytes;
    uint64_t seq;
    char name[96];
} mock_comm;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void nap(void) { struct timespec ts = {0, 100000}; nanosleep(&ts, NULL); }

int ncclGetUniqueId(ncclUniqueId* id) {
    static int counter = 0;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tsdf_mock_rccl_%d_%d_%ld", (int)getpid(), counter++, (long)time(NULL));
    return 0;
}

/* header: word 0 = joined count; slots follow: [rank][parity] = kMaxCount doubles + seq word */
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 4;   /* ncclInvalidArgument */
    mock_comm* c = (mock_comm*)calloc(1, sizeof *c);
    c->nranks = nranks; c->rank = rank;
    memcpy(c->name, id.internal, sizeof c->name - 1);
    c->bytes = 4096 + (size_t)nranks * 2 * kSlotBytes;
    const double t0 = now_s();
    int fd = -1;
    if (rank == 0) {
        shm_unlink(c->name);
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { free(c); return 2; }
    } else {
        for (;;) {
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
            if (fd >= 0) close(fd);
            if (now_s() - t0 > 20.0) { free(c); return 2; }
            nap();
        }
    }
    c->base = (char*)mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == (char*)MAP_FAILED) { free(c); return 2; }
    volatile uint64_t* joined = (volatile uint64_t*)c->base;
    __atomic_add_fetch(joined, 1, __ATOMIC_ACQ_REL);
    while (__atomic_load_n(joined, __ATOMIC_ACQUIRE) < (uint64_t)nranks) {
        if (now_s() - t0 > 20.0) { munmap(c->base, c->bytes); free(c); return 2; }
        nap();
    }
    if (rank == 0) shm_unlink(c->name);           /* everybody has it mapped: nothing stays behind */
    *comm = c;
    return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
    mock_comm* c = (mock_comm*)comm;
    if (!c || dtype != 8 /* ncclFloat64 */ || op != 0 /* ncclSum */ || count > kMaxCount) return 4;
    double mine[kMaxCount], sum[kMaxCount];
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;            /* everything queued before the collective has run */
    if (hipMemcpy(mine, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const uint64_t seq = ++c->seq;
    char* slot = c->base + 4096 + ((size_t)c->rank * 2 + (seq & 1)) * kSlotBytes;
    memcpy(slot, mine, count * sizeof(double));
    __atomic_store_n((uint64_t*)(slot + kMaxCount * sizeof(double)), seq, __ATOMIC_RELEASE);
    for (size_t e = 0; e < count; ++e) sum[e] = 0.0;
    const double t0 = now_s();
    for (int r = 0; r < c->nranks; ++r) {
        const char* s = c->base + 4096 + ((size_t)r * 2 + (seq & 1)) * kSlotBytes;
        while (__atomic_load_n((const uint64_t*)(s + kMaxCount * sizeof(double)), __ATOMIC_ACQUIRE) != seq) {
            if (now_s() - t0 > 20.0) return 6;                              /* ncclRemoteError */
        }
        for (size_t e = 0; e < count; ++e) sum[e] += ((const double*)s)[e];
    }
    if (hipMemcpy(recv, sum, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

int ncclCommDestroy(void* comm) {
    mock_comm* c = (mock_comm*)comm;
    if (c) { munmap(c->base, c->bytes); free(c); }
    return 0;
}

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case 0: return "no error";
        case 1: return "mock rccl: HIP call failed";
        case 2: return "mock rccl: shared-memory rendezvous failed";
        case 4: return "mock rccl: invalid argument";
        case 6: return "mock rccl: a rank did not reach the collective within 20 s";
        default: return "mock rccl: unknown error";
    }
}
