// api_comm.cpp -- several ranks: the in-library RCCL communicator, the shared-memory fan-in of the ranks of one node,
// the device-side peer exchange, the host hook (see handle.hpp).
#include "handle.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>

using namespace tsdf;
using namespace tsdf::host;
using namespace tsdf_api;

#include <unistd.h>

namespace tsdf_api {

void peer_close(tsdf_handle* h) {
    for (int r = 0; r < (int)h->peer.mapped.size(); ++r)
        if (h->peer.mapped[r] && h->peer.via_ipc[r]) (void)hipIpcCloseMemHandle(h->peer.mapped[r]);
    h->peer.mapped.clear();
    h->peer.via_ipc.clear();
    if (h->peer.bases_dev) (void)hipFree(h->peer.bases_dev);
    if (h->peer.own) (void)hipFree(h->peer.own);
    h->peer.bases_dev = nullptr; h->peer.own = nullptr; h->peer.nranks = 0;
}

void shm_close(tsdf_handle* h) {
    if (!h->shm.active()) return;
    (void)hipStreamSynchronize(h->stream);
    if (h->shm.dev_base) (void)hipHostUnregister(h->shm.base);
    shm_unmap(&h->shm);
}

// Shared-memory fan-in: wait for every rank's row of pass `seq`, add the leading `n` entries in rank order
// into h->red_host (the remaining entries are this rank's own).  Slots are double-buffered by pass parity: a
// rank can only overwrite its pass-s slot when publishing pass s+2, which needs everybody's pass s+1 row,

PeerExchange peer_exchange_for(const tsdf_handle* h, unsigned long long seq) {
    PeerExchange px;
    px.bases = h->peer.bases_dev;
    px.n = h->peer.nranks; px.rank = h->peer.rank;
    px.parity = (unsigned)(seq & 1ull);
    px.word = shm_word(h->shm, seq);                       // generation of the rendezvous + pass number
    px.timeout_ticks = (long long)kPeerTimeoutMs * 100000ll;   // wall_clock64(): 100 MHz
    return px;
}

}  // namespace tsdf_api

// ---- multi-GPU ---------------------------------------------------------------------------------------

int tsdf_comm_unique_id(void* id128) {
    if (!id128) return TSDF_E_BADARG;
    std::string err;
    if (!rccl::unique_id(id128, &err)) return fail(nullptr, TSDF_E_COMM, "ncclGetUniqueId: %s", err.c_str());
    return TSDF_OK;
}

int tsdf_comm_init(tsdf_handle* h, int32_t nranks, int32_t rank, const void* id128) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!id128 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(h, TSDF_E_BADARG, "tsdf_comm_init: bad argument");
    std::string err;
    if (!h->comm.init(nranks, rank, id128, &err)) return fail(h, TSDF_E_COMM, "ncclCommInitRank: %s", err.c_str());
    return TSDF_OK;
}

// The rendezvous on the named segment (host::shm_rendezvous, host_util.cpp: exclusive creation by rank 0, generation,
// unlink once everybody has joined) needs no device; what is added here is the device alias of the mapping.
int tsdf_comm_init_shm(tsdf_handle* h, int32_t nranks, int32_t rank, const char* name) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!name || nranks <= 0 || nranks > 4096 || rank < 0 || rank >= nranks) return fail(h, TSDF_E_BADARG, "tsdf_comm_init_shm: bad argument");
    peer_close(h);
    shm_close(h);
    {
        std::string serr;
        rc = shm_rendezvous(name, nranks, rank, &h->shm, &serr);
        if (rc) return fail(h, rc, "%s", serr.c_str());
    }
    char* const base = h->shm.base;
    const size_t bytes = h->shm.bytes;
    // The device alias is only needed when a rank's final kernel writes its slot itself (host fold off); with the
    // default host fold the segment is touched by hosts only, so a failed registration is not fatal.
    hipError_t e = hipHostRegister(base, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void* dptr = nullptr;
    if (e == hipSuccess) {
        e = hipHostGetDevicePointer(&dptr, base, 0);
        if (e != hipSuccess) { (void)hipHostUnregister(base); dptr = nullptr; }
    }
    if (e != hipSuccess) { (void)hipGetLastError(); dptr = nullptr; }
    h->shm.dev_base = (char*)dptr;
    h->pass_seq = 0;               // every rank counts passes from the same origin
    return TSDF_OK;
}

// Device-side exchange for the ranks of one node.  Rendezvous = tsdf_comm_init_shm (the segment carries the HIP IPC
// handles in its header and stays open); then every rank allocates its buffer (uncached device memory, zeroed),
// publishes the handle, maps everybody else's and waits until everybody has mapped everybody.
int tsdf_comm_init_peer(tsdf_handle* h, int32_t nranks, int32_t rank, const char* name) {
    if (!h) return TSDF_E_BADARG;
    if (nranks > kPeerMaxRanks) return fail(h, TSDF_E_BADARG, "tsdf_comm_init_peer: at most %d ranks (one node)", kPeerMaxRanks);
    int rc = tsdf_comm_init_shm(h, nranks, rank, name);
    if (rc) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the header keeps 64 bytes per handle");
    auto give_up = [&](int code, const char* what, hipError_t e) {
        std::string msg = std::string("tsdf_comm_init_peer: ") + what + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
        (void)hipGetLastError();
        peer_close(h);
        shm_close(h);
        return fail(h, code, "%s", msg.c_str());
    };
    const size_t bytes = (size_t)nranks * 2 * kPeerSlotBytes;
    void* own = nullptr;
    hipError_t e = hipExtMallocWithFlags(&own, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&own, bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) return give_up(TSDF_E_HIP, "no uncached / fine-grained device memory for the exchange buffer", e);
    h->peer.own = static_cast<char*>(own);
    h->peer.nranks = nranks; h->peer.rank = rank;
    h->peer.mapped.assign((size_t)nranks, nullptr);
    h->peer.via_ipc.assign((size_t)nranks, 0);
    h->peer.mapped[(size_t)rank] = h->peer.own;
    if ((e = hipMemsetAsync(own, 0, bytes, h->stream)) != hipSuccess || (e = hipStreamSynchronize(h->stream)) != hipSuccess)
        return give_up(TSDF_E_HIP, "zeroing the exchange buffer", e);
    hipIpcMemHandle_t mine;
    if ((e = hipIpcGetMemHandle(&mine, own)) != hipSuccess) return give_up(TSDF_E_COMM, "hipIpcGetMemHandle", e);
    char* entries = h->shm.base + shm_peer_entries_offset(nranks);
    auto entry_word = [&](int r, int w) { return reinterpret_cast<volatile unsigned long long*>(entries + (size_t)r * kShmPeerEntry + 64) + w; };
    const unsigned long long gen = h->shm.gen;
    std::memcpy(entries + (size_t)rank * kShmPeerEntry, &mine, sizeof mine);
    *entry_word(rank, 2) = (unsigned long long)getpid();
    *entry_word(rank, 4) = process_token();          // pids repeat across PID namespaces that share /dev/shm; this does not
    *entry_word(rank, 3) = (unsigned long long)(uintptr_t)own;
    __atomic_store_n(entry_word(rank, 0), gen, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    auto wait_for = [&](int r, int w) {
        for (unsigned spins = 0; __atomic_load_n(entry_word(r, w), __ATOMIC_ACQUIRE) != gen; ++spins) {
            if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return false;
            struct timespec ts = {0, 100000}; nanosleep(&ts, nullptr);
        }
        return true;
    };
    for (int r = 0; r < nranks; ++r) {
        if (r == rank) continue;
        if (!wait_for(r, 0)) return give_up(TSDF_E_COMM, "a rank did not publish its buffer within 20 s", hipSuccess);
        if (*entry_word(r, 2) == (unsigned long long)getpid() && *entry_word(r, 4) == process_token()) {
            // another handle of this very process: no IPC needed (or possible).  Its raw pointer is borrowed: that handle
            // must stay alive until this one has left the exchange (tsdf_comm_finalize / tsdf_destroy), see include/tsdf.h.
            h->peer.mapped[(size_t)r] = reinterpret_cast<char*>((uintptr_t)*entry_word(r, 3));
            continue;
        }
        hipIpcMemHandle_t theirs;
        std::memcpy(&theirs, entries + (size_t)r * kShmPeerEntry, sizeof theirs);
        void* ptr = nullptr;
        if ((e = hipIpcOpenMemHandle(&ptr, theirs, hipIpcMemLazyEnablePeerAccess)) != hipSuccess)
            return give_up(TSDF_E_COMM, "hipIpcOpenMemHandle", e);
        h->peer.mapped[(size_t)r] = static_cast<char*>(ptr);
        h->peer.via_ipc[(size_t)r] = 1;
    }
    if ((e = hipMalloc((void**)&h->peer.bases_dev, (size_t)nranks * sizeof(char*))) != hipSuccess ||
        (e = hipMemcpy(h->peer.bases_dev, h->peer.mapped.data(), (size_t)nranks * sizeof(char*), hipMemcpyHostToDevice)) != hipSuccess)
        return give_up(TSDF_E_HIP, "pointer table", e);
    __atomic_store_n(entry_word(rank, 1), gen, __ATOMIC_RELEASE);
    for (int r = 0; r < nranks; ++r)
        if (!wait_for(r, 1)) return give_up(TSDF_E_COMM, "a rank did not map the buffers within 20 s", hipSuccess);
    return TSDF_OK;
}

int tsdf_comm_finalize(tsdf_handle* h) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->comm.destroy();
    peer_close(h);
    shm_close(h);
    return TSDF_OK;
}

int tsdf_set_allreduce_hook(tsdf_handle* h, tsdf_allreduce_fn fn, void* ctx) {
    if (!h) return TSDF_E_BADARG;
    h->hook = fn;
    h->hook_ctx = ctx;
    return TSDF_OK;
}

int tsdf_allreduce(tsdf_handle* h, double* buf, int32_t n) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!buf || n < 0 || n > kRedWidth) return fail(h, TSDF_E_BADARG, "tsdf_allreduce: n must be in [0,%d]", kRedWidth);
    if (h->comm.active()) {
        std::string err;
        HIP_TRY(h, hipMemcpyAsync(h->red_dev, buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (!h->comm.allreduce_sum_f64(h->red_dev, n, h->stream, &err)) return fail(h, TSDF_E_COMM, "RCCL all-reduce failed: %s", err.c_str());
        HIP_TRY(h, hipMemcpyAsync(buf, h->red_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return TSDF_OK;
    }
    if (h->peer.active()) {
        // through the device, as a tracker pass does it: the row goes to every rank's buffer, the sum comes back
        const unsigned long long seq = ++h->pass_seq;
        double row[kRedWidth];
        for (int e = 0; e < kRedWidth; ++e) row[e] = e < n ? buf[e] : 0.0;
        HIP_TRY(h, hipMemcpyAsync(h->red_dev, row, sizeof row, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, launch_peer_exchange(h->stream, peer_exchange_for(h, seq), h->red_dev, n, h->red_host, seq));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->red_host[27] != h->red_host[27] && !(n > 27 && buf[27] != buf[27]))
            return fail(h, TSDF_E_COMM, "peer exchange: not every rank delivered its row within %d ms", kPeerTimeoutMs);
        std::memcpy(buf, h->red_host, (size_t)n * sizeof(double));
        return TSDF_OK;
    }
    if (h->shm.active()) {
        // host buffer in, host buffer out: publish with a host store (what a tracker pass does after its host fold)
        const unsigned long long seq = ++h->pass_seq;
        double row[kRedWidth];
        for (int e = 0; e < kRedWidth; ++e) row[e] = e < n ? buf[e] : 0.0;
        shm_publish(h->shm, seq, row);
        std::string serr;
        const int rc2 = shm_fan_in(h->shm, seq, n, h->red_host, &serr);
        if (rc2) return fail(h, rc2, "%s", serr.c_str());
        std::memcpy(buf, h->red_host, (size_t)n * sizeof(double));
        return TSDF_OK;
    }
    if (h->hook) {
        if (h->hook(buf, n, h->hook_ctx) != 0) return fail(h, TSDF_E_COMM, "all-reduce hook reported failure");
    }
    return TSDF_OK;
}
