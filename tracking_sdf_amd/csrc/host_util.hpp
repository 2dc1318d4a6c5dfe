// host_util.hpp -- the host-only pieces of the library: staging thread pool, cloud repacking, the POSIX
// shared-memory rendezvous and fan-in of the ranks of one node, slab arithmetic.  NO HIP in here: this header and
// host_util.cpp also build with plain g++ under -fsanitize=address,undefined / -fsanitize=thread
// (tests/host/host_util_test.cpp, run by pytest -m "not gpu").
#pragma once

#include "../../include/tsdf.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace tsdf {
namespace host {

// A few host threads for the one host-side job that is longer than the frame's GPU work: moving a frame handed over
// in PAGEABLE memory into the pinned staging buffers (8.3 MB of planes at 640x480, or the 19.7 MB of PCL's 32-byte
// array-of-structs points + normals they are repacked from).  run(fn) calls fn(part, parts) once per part -- part 0
// on the caller, the others on the workers -- and returns when all are done.
class HostPool {
public:
    explicit HostPool(int workers) {
        // a thread that cannot be started (resource limits) only means fewer parts: nothing may throw across the C ABI
        try {
            threads_.reserve((size_t)workers);
            for (int i = 0; i < workers; ++i) threads_.emplace_back([this, i] { loop(i + 1); });
        } catch (...) {
        }
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_.store(true, std::memory_order_release); }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int parts() const { return (int)threads_.size() + 1; }
    void run(const std::function<void(int, int)>& fn) {
        if (threads_.empty()) { fn(0, 1); return; }
        fn_ = &fn;
        pending_.store((int)threads_.size(), std::memory_order_relaxed);
        {   // the generation is published under the mutex so that a worker about to sleep cannot miss it
            std::lock_guard<std::mutex> g(mu_);
            gen_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        fn(0, parts());
        // the workers are a few microseconds behind at most: spin, then sleep
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins) {
            cpu_relax();
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                std::unique_lock<std::mutex> g(mu_);
                done_.wait(g, [this] { return pending_.load(std::memory_order_acquire) == 0; });
                break;
            }
        }
        fn_ = nullptr;
    }

private:
    static void cpu_relax() {
#if defined(__SSE2__)
        _mm_pause();
#else
        std::this_thread::yield();
#endif
    }
    // A frame's host-side work comes as 2-4 short jobs in quick succession (gather the samples, repack the cloud, repack
    // the normals, compare): waking a sleeping thread costs 20-50 us each time -- as much as the job.  A worker therefore
    // could keep looking for the next job for spin_ns_ after the last one before it sleeps.  It is 0: on the GPU boxes
    // (16-CPU quota, other tenants) 150 us of spinning changed nothing (medians 2186 against 2165 frames/s through the
    // reference's two calls, 8 alternations, profiles/r05_entry_points.json) and eleven spinning workers eat most of such a quota.
    // Round 6 tried the other end -- the hot calls telling sleeping workers to get up ahead of the job that follows
    // (profiles/r06_prewake.json): not significant for the two calls, slower for frames set one at a time.  Not kept.
    void loop(int part) {
        unsigned long long seen = 0;
        for (;;) {
            bool got = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0;; ++spins) {
                if (stop_.load(std::memory_order_acquire)) return;
                if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
                if ((spins & 63u) == 63u && std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > spin_ns_) break;
                cpu_relax();
            }
            if (!got) {
                std::unique_lock<std::mutex> g(mu_);
                sleepers_.fetch_add(1, std::memory_order_release);
                cv_.wait(g, [&] { return stop_.load(std::memory_order_acquire) || gen_.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_release);
                if (stop_.load(std::memory_order_acquire)) return;
            }
            seen = gen_.load(std::memory_order_acquire);
            const std::function<void(int, int)>* fn = fn_;
            (*fn)(part, parts());
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> g(mu_); done_.notify_one(); }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* fn_ = nullptr;
    std::atomic<unsigned long long> gen_{0};
    std::atomic<int> pending_{0}, sleepers_{0};
    std::atomic<bool> stop_{false};
    long long spin_ns_ = 0;
};


// layout of a frame block: xyz plane, nrm plane (each padded to 256 bytes), rgb plane
inline size_t plane_stride_bytes(size_t npix) { return (npix * 3 * sizeof(float) + 255) & ~(size_t)255; }
inline size_t frame_block_bytes(size_t npix) { return 2 * plane_stride_bytes(npix) + npix * 3; }

// ---- array-of-structs clouds -> planes ----------------------------------------------------------------------------
// PCL's PointXYZRGB / Normal are 32-byte structs of which 12 (+3) bytes are wanted: 19.7 MB read per 640x480 frame, 8.3 MB
// written -- the one host-side job that is longer than the frame's GPU work (2139 frames/s from PCL clouds
// against 4650 from planes).  Fast path (any layout with 16 readable bytes at the triple): four points per step, 16-byte
// loads, three shuffles, 16-byte NON-TEMPORAL stores -- the pinned planes are written once and read next by the DMA
// engine, so they need not pass through (or be read into) this core's caches.
inline void repack_triples(const char* src /* first triple */, size_t stride, bool wide /* 16 bytes readable at every triple */,
                           float* dst /* plane */, size_t i0, size_t i1) {
    size_t i = i0;
    const char* p = src + i0 * stride;
#if defined(__SSE2__)
    if (wide) {
        for (; i < i1 && (i & 3u); ++i, p += stride) std::memcpy(dst + 3 * i, p, 12);      // up to a 16-byte boundary of the plane
        for (; i + 4 <= i1; i += 4, p += 4 * stride) {
            const __m128 a = _mm_loadu_ps(reinterpret_cast<const float*>(p));
            const __m128 b = _mm_loadu_ps(reinterpret_cast<const float*>(p + stride));
            const __m128 c = _mm_loadu_ps(reinterpret_cast<const float*>(p + 2 * stride));
            const __m128 d = _mm_loadu_ps(reinterpret_cast<const float*>(p + 3 * stride));
            const __m128 t0 = _mm_shuffle_ps(a, b, _MM_SHUFFLE(0, 0, 2, 2));                // az az bx bx
            const __m128 t2 = _mm_shuffle_ps(c, d, _MM_SHUFFLE(0, 0, 2, 2));                // cz cz dx dx
            float* o = dst + 3 * i;                                                          // 16-byte aligned: i % 4 == 0, plane page-aligned
            _mm_stream_ps(o, _mm_shuffle_ps(a, t0, _MM_SHUFFLE(2, 0, 1, 0)));               // ax ay az bx
            _mm_stream_ps(o + 4, _mm_shuffle_ps(b, c, _MM_SHUFFLE(1, 0, 2, 1)));            // by bz cx cy
            _mm_stream_ps(o + 8, _mm_shuffle_ps(t2, d, _MM_SHUFFLE(2, 1, 2, 0)));           // cz dx dy dz
        }
    }
#else
    (void)wide;                                              // hosts without SSE2: the per-point copy below does all of it
#endif
    for (; i < i1; ++i, p += stride) std::memcpy(dst + 3 * i, p, 12);
}
inline void repack_aos(const tsdf_aos_layout& lay, const void* points, const void* normals, bool color,
                       float* px, float* pnm, uint8_t* pc, size_t i0, size_t i1) {
    if (points) {
        const bool wide = lay.xyz_offset + 16 <= lay.point_stride && (reinterpret_cast<uintptr_t>(px) & 15u) == 0;
        repack_triples((const char*)points + lay.xyz_offset, (size_t)lay.point_stride, wide, px, i0, i1);
        if (color) {
            const char* p = (const char*)points + i0 * (size_t)lay.point_stride;
            for (size_t i = i0; i < i1; ++i, p += lay.point_stride) {
                pc[3 * i] = (uint8_t)p[lay.r_offset]; pc[3 * i + 1] = (uint8_t)p[lay.g_offset]; pc[3 * i + 2] = (uint8_t)p[lay.b_offset];
            }
        }
    }
    if (normals) {
        const bool wide = lay.normal_offset + 16 <= lay.normal_stride && (reinterpret_cast<uintptr_t>(pnm) & 15u) == 0;
        repack_triples((const char*)normals + lay.normal_offset, (size_t)lay.normal_stride, wide, pnm, i0, i1);
    }
#if defined(__SSE2__)
    _mm_sfence();                                           // the streaming stores are globally visible before the chunk is handed to the DMA
#endif
}

// do the points [i0, i1) of an array-of-structs cloud still hold the bytes that were repacked into the planes?
inline bool points_equal_planes(const tsdf_aos_layout& lay, const void* points, bool color, const float* px, const uint8_t* pc, size_t i0, size_t i1) {
    const char* p = (const char*)points + i0 * (size_t)lay.point_stride;
    unsigned diff = 0u;
    for (size_t i = i0; i < i1; ++i, p += lay.point_stride) {
        diff |= (unsigned)(std::memcmp(px + 3 * i, p + lay.xyz_offset, 12) != 0);
        if (color) diff |= (unsigned)((uint8_t)p[lay.r_offset] ^ pc[3 * i]) | (unsigned)((uint8_t)p[lay.g_offset] ^ pc[3 * i + 1]) | (unsigned)((uint8_t)p[lay.b_offset] ^ pc[3 * i + 2]);
    }
    return diff == 0u;
}
inline bool normals_equal_plane(const tsdf_aos_layout& lay, const void* normals, const float* pnm, size_t i0, size_t i1) {
    const char* p = (const char*)normals + i0 * (size_t)lay.normal_stride + lay.normal_offset;
    unsigned diff = 0u;
    for (size_t i = i0; i < i1; ++i, p += lay.normal_stride) diff |= (unsigned)(std::memcmp(pnm + 3 * i, p, 12) != 0);
    return diff == 0u;
}

// cores this process may run on (the affinity mask: hardware_concurrency() reports the whole machine in a container)
inline int usable_cores() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return n; }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? (int)hc : 1;
}

// The tracker's sample list straight from a caller's frame (pixel p at base + p * pixel_bytes + xyz_offset: planes or
// arrays of structs), rows [r0, r1) of the sample grid, in the reference's visiting order: columns outer, rows inner
// (camera_tracking.cpp:162-163).  out: ncols x nrows entries of four floats {x, y, z, 0}.
inline void gather_samples(const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width, int32_t stride,
                           int32_t ncols, int32_t nrows, int r0, int r1, float* out) {
    for (int rj = r0; rj < r1; ++rj) {
        const char* rowp = (const char*)base + ((size_t)rj * stride * width) * pixel_bytes + xyz_offset;
        for (int ci = 0; ci < ncols; ++ci) {
            float* o = out + 4 * ((size_t)ci * nrows + rj);
            std::memcpy(o, rowp + (size_t)ci * stride * pixel_bytes, 12);
            o[3] = 0.0f;
        }
    }
}

// ---- shared-memory segment of the ranks of one node ---------------------------------------------------------------
constexpr size_t kShmSlot = 512;   // bytes per (rank, parity) slot: 34 doubles + the pass word, padded

// Segment = header, then nranks x 2 slots.  Header words (8 bytes each): magic, generation, nranks, go, joined[nranks].
constexpr unsigned long long kShmMagic = 0x5453444653484d31ull;   // "TSDFSHM1"
enum { kShmHdrMagic = 0, kShmHdrGen = 1, kShmHdrRanks = 2, kShmHdrGo = 3, kShmHdrJoined = 4 };
// Behind joined[]: one 128-byte entry per rank for tsdf_comm_init_peer (64-byte HIP IPC handle, then a word that
// turns into the generation once the handle is there, then one that does so once the rank has mapped everybody).
constexpr size_t kShmPeerEntry = 128;
inline size_t shm_peer_entries_offset(int nranks) {
    return (((size_t)kShmHdrJoined + (size_t)nranks) * 8 + kShmPeerEntry - 1) / kShmPeerEntry * kShmPeerEntry;
}
inline size_t shm_header_bytes(int nranks) {
    return (shm_peer_entries_offset(nranks) + (size_t)nranks * kShmPeerEntry + kShmSlot - 1) / kShmSlot * kShmSlot;
}
inline volatile unsigned long long* shm_hdr(char* base, int word) {
    return reinterpret_cast<volatile unsigned long long*>(base) + word;
}

struct ShmSegment {
    int nranks = 0, rank = 0;
    char* base = nullptr;        // host mapping (header, then the slots)
    char* dev_base = nullptr;    // device-visible alias (hipHostRegister; set by the caller)
    size_t bytes = 0, header = 0;
    unsigned long long gen = 0;  // generation of this segment (chosen by rank 0 at init), part of every published word
    std::string name;
    bool active() const { return base != nullptr; }
};
inline size_t shm_slot_offset(const ShmSegment& s, int rank, unsigned long long seq) {
    return s.header + ((size_t)rank * 2 + (seq & 1ull)) * kShmSlot;
}
// what a rank publishes behind its row: generation and pass number, so that a word left behind by another
// run (or another initialisation) can never be taken for this pass
inline unsigned long long shm_word(const ShmSegment& s, unsigned long long seq) {
    return (s.gen << 32) | (seq & 0xFFFFFFFFull);
}
constexpr int kShmRowDoubles = 34;           // == tsdf::kRedWidth (tsdf_device.h; checked where both are visible)

// 64 random bits drawn once per process
unsigned long long process_token();
// Rendezvous of nranks processes on the named POSIX segment (see host_util.cpp); maps it into *out.  0 or a TSDF_E_ status.
int shm_rendezvous(const char* name, int nranks, int rank, ShmSegment* out, std::string* err);
void shm_unmap(ShmSegment* s);
// this rank's row of pass `seq` into its slot (host store + release of the word)
void shm_publish(const ShmSegment& s, unsigned long long seq, const double* row /* kShmRowDoubles */);
// wait for every rank's row of pass `seq`; sum the leading n entries in rank order into red (the others: this rank's own)
int shm_fan_in(const ShmSegment& s, unsigned long long seq, int n, double* red /* kShmRowDoubles */, std::string* err);

}  // namespace host
}  // namespace tsdf
