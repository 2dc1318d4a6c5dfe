// tsdf_api.cpp -- the C ABI of include/tsdf.h: handle, device memory, frame staging, the host-side
// Gauss-Newton loop (reference src/camera_tracking.cpp:66-245) driving the HIP kernels, slab
// sharding and the per-iteration all-reduce (RCCL or a host hook).  No torch types, no exceptions
// across the boundary, no CPU fallback.
#include "../../include/tsdf.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <sched.h>
#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <ctime>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "aql_queue.hpp"
#include "host_math.hpp"
#include "rccl_dyn.hpp"
#include "tsdf_device.h"

using namespace tsdf;

namespace {

std::mutex g_err_mu;
std::string g_create_error;

struct EventPair { hipEvent_t a = nullptr, b = nullptr; };

// A few host threads for the one host-side job that is longer than the frame's GPU work: moving a frame handed over
// in PAGEABLE memory into the pinned staging buffers (8.3 MB of planes at 640x480, or the 19.7 MB of PCL's 32-byte
// array-of-structs points + normals they are repacked from).  run(fn) calls fn(part, parts) once per part -- part 0
// on the caller, the others on the workers -- and returns when all are done.
class HostPool {
public:
    explicit HostPool(int workers) {
        // a thread that cannot be started (resource limits) only means fewer parts: nothing may throw across the C ABI
        if (const char* e = std::getenv("TSDF_POOL_SPIN_US")) spin_ns_ = (long long)std::atoi(e) * 1000ll;
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
    // can keep looking for the next job for TSDF_POOL_SPIN_US microseconds after the last one before it sleeps.  Default 0:
    // on the GPU boxes (16-CPU quota, other tenants) 150 us of spinning changed nothing (medians 2186 against 2165 frames/s
    // through the reference's two calls, 8 alternations) and eleven spinning workers eat most of such a quota.
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

}  // namespace

struct tsdf_handle {
    tsdf_config cfg{};
    Grid grid{};
    hm::Pose pose{};
    double K[9]{};
    bool have_K = false;
    float v_h2_w = 0, v_h2_h = 0, v_h2_d = 0, wh2 = 0;

    int device = 0;
    hipStream_t stream = nullptr;
    float2* dw = nullptr;          // first voxel of the stored slab (inside dw_alloc, after the front padding)
    float2* dw_alloc = nullptr;
    float4* crgb = nullptr;
    int64_t n_stored = 0;          // voxels in [xs, xe)
    unsigned long long* counters = nullptr;     // device, kNumCounters
    unsigned long long* counters_host = nullptr;  // pinned
    unsigned long long* wg_counts = nullptr;      // device: {owned, halo} voxels updated, cumulative, per integrate workgroup
    unsigned long long* wg_counts_host = nullptr; // pinned mirror
    int64_t frame_serial = 0;      // frames made current so far (tsdf_frame_serial)
    // two-deep frame queue (tsdf_queue_frame / tsdf_next_frame): the NEXT frame is uploaded and packed into the pixel
    // buffer the current frame does not use while the current one is tracked and integrated
    struct Queued {
        bool active = false, direct = false, device = false, has_nrm = false, has_rgb = false;
        // device frames with deferred packing: nothing is launched when the frame is queued; the current frame's integrate
        // launch packs it (packed = true), or it becomes current unpacked like a frame of tsdf_set_frame_device
        bool deferred = false, packed = false;
        const float* d_xyz = nullptr; const float* d_nrm = nullptr; const uint8_t* d_rgb = nullptr;
        int nb = 0;
        int32_t su = 1, sv = 0;
        hipError_t err = hipSuccess;       // what the staging thread's HIP calls returned
        int rc = 0;                        // tsdf_queue_depth_frame: what the pre-processing on the staging thread returned
        std::string msg;                   // ... and its message, handed to the handle by tsdf_next_frame
    } queued;
    hipEvent_t ev_queued = nullptr;        // the queued frame's records are packed
    std::thread qthread;                   // runs the pageable path's staging so that the caller can go on tracking
    std::mutex qmu;
    std::condition_variable qcv;
    std::function<void()> qjob;
    bool qbusy = false, qstop = false;
    void* worklist = nullptr;      // integrate work items (32-byte descriptors: row << 6 | chunk, the row's share of rot_inv * g)
    unsigned* work_count = nullptr;   // work-list bookkeeping (two alternating sets: item count, band histogram, cursors)
    int integrate_blocks = 0;      // persistent grid of integrate_kernel: the most workgroups a launch uses (CUs x workgroups per CU)
    int integrate_cus = 0;         // CUs (rounded up to whole XCD groups of 8 workgroups)
    bool integrate_grid_by_work = true;   // TSDF_INTEGRATE_GRID_BY_WORK=0: always the full grid
    int integrate_debug = 0;       // timing experiments; only honoured by builds with -DTSDF_INTEGRATE_DEBUG=1

    // frame
    int32_t fw = 0, fh = 0, ncols = 0, nrows = 0, n_samples = 0;
    int32_t pix_su = 1, pix_sv = 0;   // layout of the packed pixel records of the current frame
    bool have_frame = false, frame_has_nrm = false, frame_has_rgb = false;
    float* in_xyz = nullptr; float* in_nrm = nullptr; uint8_t* in_rgb = nullptr;   // device staging (owned)
    float* pin_xyz = nullptr; float* pin_nrm = nullptr; uint8_t* pin_rgb = nullptr;  // pinned host staging (the set in use)
    // second staging set of the frame queue's pageable path: frame k+1 is filled into one set while the DMA engine still
    // reads frame k from the other (with one set the caller's thread waited for those copies before every queue call)
    float* alt_xyz = nullptr; float* alt_nrm = nullptr; uint8_t* alt_rgb = nullptr; size_t alt_cap = 0;
    // tsdf_track_aos / tsdf_integrate_aos (the reference's two calls on its own clouds): the tracker's samples go up first,
    // through their own pinned list; the cloud estimate_new_position was called with, for SDF::update's "same cloud?" check
    float4* pin_samples[2] = {nullptr, nullptr}; size_t pin_samples_cap = 0;
    struct TrackedCloud {
        bool valid = false, color = false;
        const void* points = nullptr; int32_t w = 0, h = 0; int64_t serial = -1;
        const void* normals = nullptr;         // non-null: tsdf_track_frame_aos staged the normals too (the frame is complete)
        tsdf_aos_layout lay{};
    } tracked;
    hipEvent_t ev_stage_done[2] = {nullptr, nullptr};   // [0]: the copies out of the set in use have been issued up to here; [1]: the other set's
    bool stage_recorded[2] = {false, false};
    size_t in_cap = 0;             // pixels the staging buffers hold
    bool staged_xyz = false;       // in_xyz (and in_rgb, if frame_has_rgb) hold the CURRENT frame (host / AoS / depth frames)
    std::unique_ptr<HostPool> pool;   // staging threads, started by the first pageable frame (TSDF_HOST_THREADS, default: usable cores - 2, at most 12)
    float* pre_z = nullptr; float* pre_zf = nullptr; void* pre_depth = nullptr; void* pin_depth = nullptr;   // pre-processing scratch
    size_t pre_cap = 0;
    float2* pre_grid_a = nullptr; float2* pre_grid_b = nullptr; size_t pre_grid_cap = 0;                     // bilateral grid (cells)
    unsigned* pre_minmax = nullptr; unsigned* pin_minmax = nullptr;                                           // depth range words
    float4* pn = nullptr;          // 2 x float4 per pixel      } the CURRENT frame's buffers: one of the two below
    float4* samples = nullptr;     //                            }
    size_t pn_cap = 0, samples_cap = 0;
    // Frame side stream: H2D staging copies, pre-processing and pack_kernel of frame k+1 run on `fstream`, so they
    // overlap the integration of frame k that is still running on `stream`.  The packed records are double-buffered;
    // `ev_frame` makes `stream` wait for the pack, `ev_buf_used[b]` makes the pack wait for the last integration that
    // read buffer b (the tracker passes are host-synchronous and need no event).
    hipStream_t fstream = nullptr;
    hipEvent_t ev_frame = nullptr;
    hipEvent_t ev_samples = nullptr;           // the frame's sample list is on the device (samples-first uploads)
    bool records_pending = false;              // the frame's pixel records are still being produced on the frame stream (ev_frame):
                                               // the tracker may run (it reads the sample list), tsdf_integrate waits for them
    hipEvent_t ev_copied = nullptr;            // the H2D copies of a frame handed over in page-locked caller buffers
    hipEvent_t ev_buf_used[2] = {nullptr, nullptr};
    bool used_valid[2] = {false, false};
    bool used_untracked[2] = {false, false};   // read by an integration that recorded no event
    bool frame_side = false;                   // the current frame was packed on the frame stream
    // the current frame's records (pn) and sample list are still to be written: tsdf_set_frame_device leaves the
    // packing to the integrate launch, and the tracker reads the samples from the xyz plane meanwhile (defer_pack)
    struct DeferredPack {
        bool pending = false, samples_listed = false;      // samples_listed: a tracker pass has written the sample list
        const float* xyz = nullptr; const float* nrm = nullptr; const uint8_t* rgb = nullptr;
    } deferred;
    // tsdf_device_frame_released: frames handed over in DEVICE memory whose planes the library may still read, oldest
    // first.  A frame is free once the launch that packs it has run: that launch stores its ticket into release_host[s]
    // (pinned; s = 0 main stream, 1 frame stream -- tickets grow per stream) and the entry says which ticket to wait for.
    struct BorrowedFrame {
        int64_t serial;                        // tsdf_frame_serial() the frame has (or will have, while it is still queued)
        int stream;                            // -1: not packed yet (and not abandoned): still borrowed, whatever the words say
        unsigned long long ticket;
    };
    std::deque<BorrowedFrame> borrowed;
    int64_t borrow_lost = -1;                      // >= 0: an entry could not be recorded for this serial (see borrow_device_frame)
    unsigned long long* release_host = nullptr;    // pinned: [0], [1] the two streams' tickets; [2] work items of the last integrate launch
    unsigned long long release_ticket[2] = {0ull, 0ull};
    bool deferred_list_samples = true;         // TSDF_DEFER_PACK=2 (diagnosis): every pass reads the plane
    bool defer_device_pack = true;             // TSDF_DEFER_PACK=0: pack when the frame is set
    float4* pn_buf[2] = {nullptr, nullptr};
    bool integrate_queue = false;  // integrate_queue_kernel (dense batches, TSDF_INTEGRATE_KERNEL=queue) rather than integrate_kernel
    float4* samples_buf[2] = {nullptr, nullptr};
    int fidx = 0;

    // tracker reduction buffers
    double* partials = nullptr; size_t partials_cap = 0;   // doubles
    double* red_dev = nullptr;     // kRedWidth
    double* red_host = nullptr;    // pinned, kRedWidth doubles + the pass-number word the host polls
    unsigned long long pass_seq = 0;
    bool host_fold = true;         // shared-memory fan-in: the host publishes this rank's row (TSDF_HOST_FOLD=0: the device writes the slot itself)
    unsigned* fold_ctr = nullptr;  // arrival counters of the in-launch fan-in of track_kernel: two sets, alternating by pass parity
    // Gauss-Newton passes >= 1 of a tsdf_track call go through a user-mode queue of the library's own (2.3 us per pass less
    // than hipLaunchKernel: profiles/r05_aql_probe.json) when the row is handed to this host anyway (single rank / host
    // fan-in); pass 0 stays on the stream, behind the integration.  TSDF_AQL=0: every pass through the stream.
    AqlQueue aql;
    bool aql_on = false;
    long long aql_passes = 0;
    double* shard_host = nullptr;  // pinned: kTrackShards slots of kShardSlotDoubles (host side of the fan-in)
    bool host_fanin = true;        // the second level of the tracker fan-in runs on the host (TSDF_HOST_FANIN=0: on the device)
    unsigned integrate_launches = 0;

    bool poll = true;              // spin on the pass-number word instead of hipStreamSynchronize
    unsigned long long* track_stamps = nullptr;   // TSDF_TRACK_STAMPS=1: 8 device words per tracker workgroup (phase stamps of the last pass)
    // TSDF_TRACK_PROFILE=1: host-side clock of a pass, printed by tsdf_destroy (ns sums: parameters, launch call, wait
    // for the row, fold + solve + pose)
    bool track_profile = false;
    double tp_fill = 0, tp_launch = 0, tp_wait = 0, tp_post = 0; long long tp_passes = 0;
    // TSDF_STAGE_PROFILE=1: host-side clock of the pageable-frame staging (ns sums per staged frame, printed by tsdf_destroy)
    struct StageProfile {
        bool on = false;
        long long frames = 0;
        double total = 0;        // stage_and_upload, first call to return
        double fill_max = 0;     // the slowest worker's filling time (sum over chunks)
        double first_chunk = 0;  // until the first chunk was complete
        double upload_calls = 0; // inside hipMemcpyAsync
        double sync_before = 0;  // queue_frame: waiting for the previous frame's copies to leave the staging planes
        double next_wait = 0;    // tsdf_next_frame: waiting for the staging thread
        double handoff = 0;      // queue call -> the staging thread starts the job
        // tsdf_track_aos / tsdf_integrate_aos (ns sums)
        long long aos_frames = 0;
        double a_prep1 = 0, a_prep2 = 0, a_prep = 0, a_gather = 0, a_issue = 0, a_loop = 0, a_wait = 0, b_normals = 0, b_verify = 0, b_issue = 0, b_integrate = 0;
    } sp;

    // comm
    rccl::Comm comm;
    tsdf_allreduce_fn hook = nullptr;
    void* hook_ctx = nullptr;
    // shared-memory fan-in (ranks of one node): nranks x 2 slots (double-buffered by pass parity)
    struct Shm {
        int nranks = 0, rank = 0;
        char* base = nullptr;        // host mapping (header, then the slots)
        char* dev_base = nullptr;    // device-visible alias (hipHostRegister)
        size_t bytes = 0, header = 0;
        unsigned long long gen = 0;  // generation of this segment (chosen by rank 0 at init), part of every published word
        std::string name;
        bool active() const { return base != nullptr; }
    } shm;

    // device-side exchange between the ranks of one node (tsdf_comm_init_peer); the shared segment above stays open
    // next to it: it carried the IPC handles
    struct Peer {
        int nranks = 0, rank = 0;
        char* own = nullptr;            // this rank's buffer: nranks x 2 slots of kPeerSlotBytes, uncached device memory
        std::vector<char*> mapped;      // rank r's buffer as mapped here (mapped[rank] == own)
        std::vector<char> via_ipc;      // mapped[r] came from hipIpcOpenMemHandle (and is closed again); a sibling handle of
                                        // this process lends its raw pointer instead and must outlive this handle's exchange
        char** bases_dev = nullptr;     // the same pointers on the device
        bool active() const { return own != nullptr; }
    } peer;

    // tsdf_sample scratch (grown on demand, kept between calls)
    double* sample_vox = nullptr; float* sample_val = nullptr; int32_t* sample_ok = nullptr; size_t sample_cap = 0;

    // mesh extraction (grown on demand, kept between calls)
    unsigned* mesh_row_count = nullptr; unsigned* mesh_row_offset = nullptr; size_t mesh_rows_cap = 0;
    unsigned* mesh_group_sum = nullptr; unsigned long long* mesh_group_base = nullptr;   // per 1024 rows
    unsigned long long* mesh_total = nullptr;     // pinned: triangles of the last count pass, then the violation word
    float* mesh_verts = nullptr; float4* mesh_colors = nullptr; unsigned long long* mesh_desc = nullptr;
    size_t mesh_verts_cap = 0, mesh_colors_cap = 0;   // triangles
    int64_t mesh_ntri = -1;                        // -1: nothing extracted yet
    bool mesh_has_color = false;

    // measurement
    bool timing = false;           // events around the integrate / pack launches (asynchronous, drained on read)
    int timing_period = 1;         // 1 = every launch, n = every n-th launch of a kind
    unsigned timing_seen[2] = {0u, 0u};
    bool timing_track = false;     // events around every tracker pass (needs a completed stop event per pass)
    std::vector<EventPair> ev_pool;   // pending integrate/pack pairs
    std::vector<int> ev_kind;         // 0 = integrate, 1 = pack
    size_t ev_used = 0;
    EventPair ev_track;
    tsdf_timing tm{};
    tsdf_counters cnt{};
    unsigned long long cnt_base[kNumCounters]{};

    std::string err;
};

namespace {

// Where fail() leaves its message when it runs on the queue's library thread (tsdf_queue_depth_frame): the handle's
// own string belongs to the caller's thread, which may be failing a call of its own at that moment.
thread_local std::string* t_err_sink = nullptr;

int fail(tsdf_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h && t_err_sink) *t_err_sink = buf;
    else if (h) h->err = buf;
    else { std::lock_guard<std::mutex> lk(g_err_mu); g_create_error = buf; }
    return code;
}

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail((h), TSDF_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

int bind_device(tsdf_handle* h) {
    HIP_TRY(h, hipSetDevice(h->device));
    return TSDF_OK;
}

void free_preproc(tsdf_handle* h) {
    if (h->pre_z) (void)hipFree(h->pre_z);
    if (h->pre_zf) (void)hipFree(h->pre_zf);
    if (h->pre_depth) (void)hipFree(h->pre_depth);
    if (h->pin_depth) (void)hipHostFree(h->pin_depth);
    if (h->pre_grid_a) (void)hipFree(h->pre_grid_a);
    if (h->pre_grid_b) (void)hipFree(h->pre_grid_b);
    if (h->pre_minmax) (void)hipFree(h->pre_minmax);
    if (h->pin_minmax) (void)hipHostFree(h->pin_minmax);
    h->pre_z = h->pre_zf = nullptr; h->pre_depth = h->pin_depth = nullptr; h->pre_cap = 0;
    h->pre_grid_a = h->pre_grid_b = nullptr; h->pre_grid_cap = 0;
    h->pre_minmax = h->pin_minmax = nullptr;
}

// layout of a frame block: xyz plane, nrm plane (each padded to 256 bytes), rgb plane
inline size_t plane_stride_bytes(size_t npix) { return (npix * 3 * sizeof(float) + 255) & ~(size_t)255; }
inline size_t frame_block_bytes(size_t npix) { return 2 * plane_stride_bytes(npix) + npix * 3; }

void free_frame(tsdf_handle* h) {
    // (xyz | nrm | rgb live in ONE block each: the xyz pointer is the block)
    if (h->in_xyz) (void)hipFree(h->in_xyz);
    if (h->pin_xyz) (void)hipHostFree(h->pin_xyz);
    if (h->alt_xyz) (void)hipHostFree(h->alt_xyz);
    for (int b = 0; b < 2; ++b) { if (h->pin_samples[b]) (void)hipHostFree(h->pin_samples[b]); h->pin_samples[b] = nullptr; }
    h->pin_samples_cap = 0;
    h->tracked = tsdf_handle::TrackedCloud();
    h->in_xyz = h->in_nrm = nullptr; h->in_rgb = nullptr;
    h->pin_xyz = h->pin_nrm = nullptr; h->pin_rgb = nullptr;
    h->alt_xyz = h->alt_nrm = nullptr; h->alt_rgb = nullptr; h->alt_cap = 0;
    h->stage_recorded[0] = h->stage_recorded[1] = false;
    h->in_cap = 0;
}

int ensure_frame_buffers(tsdf_handle* h, int32_t w, int32_t hh, bool need_staging) {
    const size_t npix = (size_t)w * hh;
    const int32_t st = h->cfg.pixel_stride;
    const int32_t ncols = (w + st - 1) / st, nrows = (hh + st - 1) / st;
    const size_t ns = (size_t)ncols * nrows;
    if (npix > h->pn_cap || ns > h->samples_cap) {
        // growing: nothing may still read the old buffers
        HIP_TRY(h, hipStreamSynchronize(h->fstream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->used_valid[0] = h->used_valid[1] = false;
        h->used_untracked[0] = h->used_untracked[1] = false;
    }
    if (npix > h->pn_cap) {
        for (int b = 0; b < 2; ++b) { if (h->pn_buf[b]) (void)hipFree(h->pn_buf[b]); h->pn_buf[b] = nullptr; }
        h->pn = nullptr; h->pn_cap = 0; h->have_frame = false;
        for (int b = 0; b < 2; ++b) HIP_TRY(h, hipMalloc((void**)&h->pn_buf[b], npix * kPixelBufferBytes));
        h->pn_cap = npix;
    }
    if (ns > h->samples_cap) {
        for (int b = 0; b < 2; ++b) { if (h->samples_buf[b]) (void)hipFree(h->samples_buf[b]); h->samples_buf[b] = nullptr; }
        h->samples = nullptr; h->samples_cap = 0; h->have_frame = false;
        for (int b = 0; b < 2; ++b) HIP_TRY(h, hipMalloc((void**)&h->samples_buf[b], ns * sizeof(float4)));
        h->samples_cap = ns;
    }
    const size_t nb = track_partials_doubles((int32_t)ns);
    if (nb > h->partials_cap) {
        if (h->partials) (void)hipFree(h->partials);
        h->partials = nullptr; h->partials_cap = 0;
        HIP_TRY(h, hipMalloc((void**)&h->partials, nb * sizeof(double)));
        h->partials_cap = nb;
    }
    if (need_staging && npix > h->in_cap) {
        free_frame(h);
        // the three planes of a frame in ONE block, on the device and in the pinned staging set alike: a staged frame is
        // then ONE host-to-device copy (measured: the copies of a 640x480 frame are bound by their number, not their
        // bytes -- 12 copies per frame 3450 frames/s from PCL clouds, 3 copies 4140)
        const size_t plane = plane_stride_bytes(npix);
        char* dev = nullptr; char* pin = nullptr;
        HIP_TRY(h, hipMalloc((void**)&dev, frame_block_bytes(npix)));
        h->in_xyz = reinterpret_cast<float*>(dev); h->in_nrm = reinterpret_cast<float*>(dev + plane); h->in_rgb = reinterpret_cast<uint8_t*>(dev + 2 * plane);
        HIP_TRY(h, hipHostMalloc((void**)&pin, frame_block_bytes(npix), hipHostMallocDefault));
        h->pin_xyz = reinterpret_cast<float*>(pin); h->pin_nrm = reinterpret_cast<float*>(pin + plane); h->pin_rgb = reinterpret_cast<uint8_t*>(pin + 2 * plane);
        h->in_cap = npix;
    }
    h->fw = w; h->fh = hh; h->ncols = ncols; h->nrows = nrows; h->n_samples = (int32_t)ns;
    return TSDF_OK;
}

// ---- event timing -------------------------------------------------------------------------------

int drain_events(tsdf_handle* h) {
    if (h->ev_used == 0) return TSDF_OK;
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b));
        if (h->ev_kind[i] == 0) { h->tm.integrate_ms += ms; h->tm.integrate_launches++; }
        else { h->tm.pack_ms += ms; h->tm.pack_launches++; }
    }
    h->ev_used = 0;
    return TSDF_OK;
}

int timed_begin(tsdf_handle* h, int kind, EventPair** out, hipStream_t st) {
    *out = nullptr;
    if (!h->timing) return TSDF_OK;
    // sampling: an event pair around every launch costs the frame loop several microseconds per pair (measured: 6 % of
    // the frame rate at 512^3), so only every timing_period-th launch of a kind is bracketed
    if (h->timing_period > 1 && (h->timing_seen[kind & 1]++ % (unsigned)h->timing_period) != 0) return TSDF_OK;
    if (h->ev_used == h->ev_pool.size()) {
        if (h->ev_pool.size() >= 4096) {
            int rc = drain_events(h);
            if (rc) return rc;
        } else {
            EventPair ep;
            HIP_TRY(h, hipEventCreate(&ep.a));
            HIP_TRY(h, hipEventCreate(&ep.b));
            h->ev_pool.push_back(ep);
            h->ev_kind.push_back(0);
        }
    }
    EventPair* ep = &h->ev_pool[h->ev_used];
    h->ev_kind[h->ev_used] = kind;
    h->ev_used++;
    HIP_TRY(h, hipEventRecord(ep->a, st));
    *out = ep;
    return TSDF_OK;
}

int timed_end(tsdf_handle* h, EventPair* ep, hipStream_t st) {
    if (ep) HIP_TRY(h, hipEventRecord(ep->b, st));
    return TSDF_OK;
}

// Record layout for this frame: along one voxel k-row the projection moves by
// d(u,v)/dk ~ (K row 0 . c, K row 1 . c) with c = third column of rot_inv (evaluated on the optical
// axis).  If it moves mostly down the image, store the records column-major so that the gather of 64
// consecutive k reads neighbouring records; otherwise row-major.  Results do not depend on it.
void pick_pixel_layout(const tsdf_handle* h, int32_t* su, int32_t* sv) {
    const double* Ri = h->pose.rot_inv;
    const double du = h->have_K ? h->K[0] * Ri[2] + h->K[1] * Ri[5] : Ri[2];
    const double dv = h->have_K ? h->K[3] * Ri[2] + h->K[4] * Ri[5] : Ri[5];
    if (std::fabs(dv) >= std::fabs(du)) { *su = h->fh; *sv = 1; }
    else { *su = 1; *sv = h->fw; }
}
void choose_pixel_layout(tsdf_handle* h) { pick_pixel_layout(h, &h->pix_su, &h->pix_sv); }

// make stream `st` wait until the integration that last read pixel buffer nb is done
int wait_buffer_free(tsdf_handle* h, int nb, hipStream_t st) {
    if (h->used_valid[nb]) {
        HIP_TRY(h, hipStreamWaitEvent(st, h->ev_buf_used[nb], 0));
    } else if (h->used_untracked[nb]) {
        // the buffer was last read by an integration that recorded no event (device-resident frames do not pay
        // for one): order behind everything queued on the main stream, once
        HIP_TRY(h, hipEventRecord(h->ev_buf_used[nb], h->stream));
        HIP_TRY(h, hipStreamWaitEvent(st, h->ev_buf_used[nb], 0));
    }
    h->used_untracked[nb] = false;
    return TSDF_OK;
}

PackArgs pack_args(const tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t su, int32_t sv, int nb) {
    PackArgs a;
    a.xyz = xyz; a.nrm = nrm; a.rgb = rgb;
    a.width = h->fw; a.height = h->fh; a.stride = h->cfg.pixel_stride;
    a.pix_su = su; a.pix_sv = sv;
    a.pn = h->pn_buf[nb]; a.samples = h->samples_buf[nb];
    a.ncols = h->ncols; a.nrows = h->nrows;
    a.color_layout = h->cfg.with_color ? (h->integrate_queue ? 2 : 1) : 0;
    return a;
}

// ---- borrowed device planes (tsdf_device_frame_released) ---------------------------------------------------------
// a device frame with this serial has been handed over; nothing has packed it yet
void borrow_device_frame(tsdf_handle* h, int64_t serial) {
    try { h->borrowed.push_back({serial, -1, 0ull}); }
    catch (...) { if (h->borrow_lost < 0) h->borrow_lost = serial; }    // out of memory for 24 bytes: nothing may throw across the C ABI;
}                                                                        // frames from here on are reported borrowed until tsdf_synchronize
// the launch that packs frame `serial` is about to be issued on stream index s: the ticket it will publish
ReleaseWord release_for(tsdf_handle* h, int64_t serial, int s) {
    ReleaseWord r;
    r.word = h->release_host + s;
    r.ticket = ++h->release_ticket[s];
    for (auto& b : h->borrowed)
        if (b.serial == serial) { b.stream = s; b.ticket = r.ticket; }
    return r;
}
// frame `serial` will never be packed (replaced while its packing was still deferred; the tracker passes that read its
// xyz plane are host-synchronous and over): free as soon as the frames before it are
void abandon_device_frame(tsdf_handle* h, int64_t serial) {
    for (auto& b : h->borrowed)
        if (b.serial == serial && b.stream < 0) { b.stream = 0; b.ticket = 0ull; }
}
// newest serial S such that no device frame with serial <= S is still read by the library
int64_t released_serial(tsdf_handle* h) {
    while (!h->borrowed.empty()) {
        const tsdf_handle::BorrowedFrame& b = h->borrowed.front();
        if (b.stream < 0) break;
        if (b.ticket && __atomic_load_n(h->release_host + b.stream, __ATOMIC_ACQUIRE) < b.ticket) break;
        h->borrowed.pop_front();
    }
    int64_t rel = h->borrowed.empty() ? h->frame_serial + ((h->queued.active && h->queued.device) ? 1 : 0) : h->borrowed.front().serial - 1;
    if (h->borrow_lost >= 0 && rel >= h->borrow_lost) rel = h->borrow_lost - 1;
    return rel;
}

// st = h->fstream when the inputs were produced on the frame stream (host images, pre-processing): the pack then
// overlaps the running integration like they do.  Device-resident inputs pack on the main stream: measured, a
// pack_kernel squeezed in beside the persistent integrate_kernel slows that one down by as much as it takes.
int run_pack(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, hipStream_t st, bool borrowed_planes = false,
             bool samples_first = false /* the sample list of this frame went up ahead (upload_samples_first): the pack writes the
                                           records only and the main stream is NOT made to wait for it here -- tsdf_integrate does */) {
    if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);      // the frame this one replaces was never packed
    choose_pixel_layout(h);
    const int nb = h->fidx ^ 1;                               // the buffer the previous frame did not use
    const bool side = st != h->stream;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first");
    if (side) { const int rcw = wait_buffer_free(h, nb, st); if (rcw) return rcw; }
    h->frame_side = side;
    EventPair* ep;
    int rc = timed_begin(h, 1, &ep, st);
    if (rc) return rc;
    {
        PackArgs pa = pack_args(h, xyz, nrm, rgb, h->pix_su, h->pix_sv, nb);
        if (samples_first) pa.samples = nullptr;
        HIP_TRY(h, launch_pack(st, pa));
    }
    if (borrowed_planes) {
        borrow_device_frame(h, h->frame_serial + 1);
        HIP_TRY(h, launch_release(st, release_for(h, h->frame_serial + 1, side ? 1 : 0)));
    }
    rc = timed_end(h, ep, st);
    if (rc) return rc;
    h->records_pending = false;
    if (side) {
        HIP_TRY(h, hipEventRecord(h->ev_frame, st));
        if (samples_first) h->records_pending = true;          // the tracker needs the sample list only (main stream waits for ev_samples)
        else HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_frame, 0));   // everything queued on `stream` from here on sees the frame
    }
    h->fidx = nb; h->pn = h->pn_buf[nb]; h->samples = h->samples_buf[nb];
    h->have_frame = true;
    h->frame_serial++;
    h->frame_has_nrm = nrm != nullptr;
    h->frame_has_rgb = rgb != nullptr;
    h->deferred = tsdf_handle::DeferredPack();
    return TSDF_OK;
}

// Samples first: a frame that arrives in host memory needs 8.3 MB (640x480) on the device before it can be integrated, but the
// tracker only reads every pixel_stride-th point of every pixel_stride-th row -- 34 240 points, 0.5 MB.  They are gathered
// from the caller's memory (pixel p at base + p * pixel_bytes + xyz_offset: planes or arrays of structs) by the staging
// threads, in the reference's visiting order (columns outer, rows inner, camera_tracking.cpp:162-163), copied in front of
// everything else of the frame, and the main stream waits for THAT copy only: the Gauss-Newton passes run while the planes
// are still travelling.  Writes the list of the record buffer the frame is about to take (fidx ^ 1).
HostPool* host_pool(tsdf_handle* h);        // (the staging threads, defined with the frame entry points below)
int ensure_pin_samples(tsdf_handle* h) {
    const size_t ns = (size_t)h->n_samples;
    if (ns <= h->pin_samples_cap) return TSDF_OK;
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    for (int b = 0; b < 2; ++b) { if (h->pin_samples[b]) (void)hipHostFree(h->pin_samples[b]); h->pin_samples[b] = nullptr; }
    h->pin_samples_cap = 0;
    for (int b = 0; b < 2; ++b) HIP_TRY(h, hipHostMalloc((void**)&h->pin_samples[b], ns * sizeof(float4), hipHostMallocDefault));
    h->pin_samples_cap = ns;
    return TSDF_OK;
}
int upload_samples_first(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width) {
    int rc = ensure_pin_samples(h);
    if (rc) return rc;
    const int nb = h->fidx ^ 1;
    float4* const ps = h->pin_samples[nb];
    const int32_t st = h->cfg.pixel_stride, ncols = h->ncols, nrows = h->nrows;
    const std::function<void(int, int)> gather = [&](int part, int parts) {
        const int r0 = (int)((long long)nrows * part / parts), r1 = (int)((long long)nrows * (part + 1) / parts);
        for (int rj = r0; rj < r1; ++rj) {
            const char* rowp = (const char*)base + ((size_t)rj * st * width) * pixel_bytes + xyz_offset;
            for (int ci = 0; ci < ncols; ++ci) {
                float4 v;
                std::memcpy(&v, rowp + (size_t)ci * st * pixel_bytes, 12);
                v.w = 0.0f;
                ps[(size_t)ci * nrows + rj] = v;
            }
        }
    };
    HostPool* const pool = host_pool(h);
    if (pool) pool->run(gather); else gather(0, 1);
    HIP_TRY(h, hipMemcpyAsync(h->samples_buf[nb], ps, (size_t)h->n_samples * sizeof(float4), hipMemcpyHostToDevice, h->fstream));
    HIP_TRY(h, hipEventRecord(h->ev_samples, h->fstream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_samples, 0));
    return TSDF_OK;
}

// A frame handed over in device memory is not packed when it is set: the tracker reads its samples from the xyz plane
// (TrackParams::xyz_plane) and the pixel records are written inside the integrate launch, by workgroups appended to
// list_rows_kernel (launch_integrate) -- the packing then hides under that kernel's latency chain instead of being 11 us
// of its own in front of the first tracker pass.  The planes stay borrowed until that launch has run: tsdf.h asks for them
// until tsdf_device_frame_released() reaches the frame's serial (or tsdf_synchronize, which packs what is pending).
// TSDF_DEFER_PACK=0: pack at once, as rounds 1-3 did.
int defer_pack(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, bool already_borrowed = false) {
    if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);      // the frame this one replaces was never packed
    if (!already_borrowed) borrow_device_frame(h, h->frame_serial + 1);
    choose_pixel_layout(h);
    const int nb = h->fidx ^ 1;
    h->frame_side = false;
    h->fidx = nb; h->pn = h->pn_buf[nb]; h->samples = h->samples_buf[nb];
    h->have_frame = true;
    h->frame_serial++;
    h->frame_has_nrm = nrm != nullptr;
    h->frame_has_rgb = rgb != nullptr;
    h->deferred = tsdf_handle::DeferredPack();
    h->deferred.pending = true; h->deferred.xyz = xyz; h->deferred.nrm = nrm; h->deferred.rgb = rgb;
    h->records_pending = false;
    return TSDF_OK;
}

void fill_integrate_params(const tsdf_handle* h, IntegrateParams& p) {
    p.g = h->grid;
    std::memcpy(p.rot_inv, h->pose.rot_inv, sizeof p.rot_inv);
    std::memcpy(p.rot_inv_trans, h->pose.rot_inv_trans, sizeof p.rot_inv_trans);
    std::memcpy(p.K, h->K, sizeof p.K);
    p.width = h->fw; p.height = h->fh;
    p.pix_su = h->pix_su; p.pix_sv = h->pix_sv;
    p.with_color = h->cfg.with_color;
    p.debug = h->integrate_debug;
}

void fill_track_params(const tsdf_handle* h, TrackParams& p) {
    p.g = h->grid;
    std::memcpy(p.rot, h->pose.rot, sizeof p.rot);
    std::memcpy(p.trans, h->pose.trans, sizeof p.trans);
    hm::perturbed_rotations(h->pose, h->cfg.w_h, p.rpm);
    p.v_h = h->cfg.v_h;
    p.vh2[0] = h->v_h2_w; p.vh2[1] = h->v_h2_h; p.vh2[2] = h->v_h2_d;
    p.wh2 = h->wh2;
    p.n_samples = h->n_samples;
    p.stale_carry = h->cfg.stale_carry;
    p.carry_threads = h->cfg.carry_threads > 1 ? h->cfg.carry_threads : 1;
    p.ncols = h->ncols; p.nrows = h->nrows;
}

constexpr size_t kVolumePadFront = 16;   // voxels of {0,0} padding in front of the volume (128 bytes)
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
inline size_t shm_slot_offset(const tsdf_handle* h, int rank, unsigned long long seq) {
    return h->shm.header + ((size_t)rank * 2 + (seq & 1ull)) * kShmSlot;
}
// what a rank publishes behind its row: generation and pass number, so that a word left behind by another
// run (or another initialisation) can never be taken for this pass
inline unsigned long long shm_word(const tsdf_handle* h, unsigned long long seq) {
    return (h->shm.gen << 32) | (seq & 0xFFFFFFFFull);
}

// 64 random bits drawn once per process: tells "another handle of this process" from "a process with the same pid in
// another PID namespace" when peers compare notes in the shared segment
unsigned long long process_token() {
    static const unsigned long long tok = [] {
        unsigned long long t = 0;
        if (FILE* f = std::fopen("/dev/urandom", "rb")) { if (std::fread(&t, sizeof t, 1, f) != 1) t = 0; std::fclose(f); }
        if (!t) t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 32) ^ (unsigned long long)(uintptr_t)&t;
        return t | 1ull;
    }();
    return tok;
}

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
    munmap(h->shm.base, h->shm.bytes);
    // the name was unlinked by rank 0 as soon as every rank had joined (tsdf_comm_init_shm): nothing to remove
    // here, and by now the name may belong to somebody else
    h->shm.base = h->shm.dev_base = nullptr;
    h->shm.nranks = 0;
}

// Shared-memory fan-in: wait for every rank's row of pass `seq`, add the leading `n` entries in rank order
// into h->red_host (the remaining entries are this rank's own).  Slots are double-buffered by pass parity: a
// rank can only overwrite its pass-s slot when publishing pass s+2, which needs everybody's pass s+1 row,
// which nobody publishes before having read all pass-s rows.
int shm_fan_in(tsdf_handle* h, unsigned long long seq, int n) {
    double sum[kRedWidth];
    for (int e = 0; e < kRedWidth; ++e) sum[e] = 0.0;
    const unsigned long long want = shm_word(h, seq);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < h->shm.nranks; ++r) {
        const char* slot = h->shm.base + shm_slot_offset(h, r, seq);
        const volatile unsigned long long* word =
            reinterpret_cast<const volatile unsigned long long*>(slot + kRedWidth * sizeof(double));
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) break;
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
                return fail(h, TSDF_E_COMM, "shared-memory fan-in: rank %d did not publish pass %llu within 20 s", r, seq);
        }
        const double* row = reinterpret_cast<const double*>(slot);
        for (int e = 0; e < n; ++e) sum[e] += row[e];
        if (r == h->shm.rank) std::memcpy(h->red_host, row, kRedWidth * sizeof(double));
    }
    std::memcpy(h->red_host, sum, (size_t)n * sizeof(double));
    return TSDF_OK;
}

constexpr int kPeerTimeoutMs = 5000;
PeerExchange peer_exchange_for(const tsdf_handle* h, unsigned long long seq) {
    PeerExchange px;
    px.bases = h->peer.bases_dev;
    px.n = h->peer.nranks; px.rank = h->peer.rank;
    px.parity = (unsigned)(seq & 1ull);
    px.word = shm_word(h, seq);                       // generation of the rendezvous + pass number
    px.timeout_ticks = (long long)kPeerTimeoutMs * 100000ll;   // wall_clock64(): 100 MHz
    return px;
}

// Launch one accumulation pass and wait for its kRedWidth-double result row in h->red_host.
// reduce_ranks: sum the leading kRedAllreduce entries over ranks (RCCL on the device buffer, or hook).
int accumulate_pass(tsdf_handle* h, bool reduce_ranks, bool later_pass = false /* pass >= 1 of a tsdf_track call */) {
    const auto tp0 = h->track_profile ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
    TrackParams p;
    fill_track_params(h, p);
    // deferred packing: the first pass over the frame reads its samples from the xyz plane and leaves them in the list
    const bool from_plane = h->deferred.pending && !h->deferred.samples_listed;
    if (from_plane) {
        p.xyz_plane = h->deferred.xyz; p.plane_width = h->fw; p.pixel_stride = h->cfg.pixel_stride;
        p.sample_list_out = h->samples;
    }
    const bool use_rccl = reduce_ranks && h->comm.active();
    const bool use_peer = reduce_ranks && !use_rccl && h->peer.active();
    const bool use_shm = reduce_ranks && !use_rccl && !use_peer && h->shm.active();
    const unsigned long long seq = ++h->pass_seq;
    // Where the last workgroup of track_kernel publishes this rank's finished row: pinned host memory (the host then
    // also publishes it into the shared segment with a host store), or -- TSDF_HOST_FOLD=0 -- straight into this rank's
    // slot of the shared segment through its device alias, tagged with the generation the other ranks wait for.
    const bool dev_publish_shm = use_shm && !h->host_fold;
    if (dev_publish_shm && !h->shm.dev_base)
        return fail(h, TSDF_E_COMM, "shared-memory fan-in with device publishing needs the segment registered with HIP, which failed");
    double* host_row = dev_publish_shm
        ? reinterpret_cast<double*>(h->shm.dev_base + shm_slot_offset(h, h->shm.rank, seq)) : h->red_host;
    const unsigned long long dev_word = dev_publish_shm ? shm_word(h, seq) : seq;
    if (h->timing_track) HIP_TRY(h, hipEventRecord(h->ev_track.a, h->stream));
    // the row ends up on this host anyway (no in-stream all-reduce, no device-published slot): let the device stop
    // after the shard level and add the <= 8 shard rows here
    const bool host_fanin = h->host_fanin && h->poll && !use_rccl && !use_peer && !dev_publish_shm && !h->timing_track;
    // device-side exchange: the workgroup that finishes the row swaps it with the other ranks before handing it out
    PeerExchange px;
    if (use_peer) px = peer_exchange_for(h, seq);
    const auto tp1 = h->track_profile ? std::chrono::steady_clock::now() : tp0;
    // passes >= 1 whose row comes to this host: through the library's own queue (pass 0 stays on the stream, ordered behind
    // the integration; a later pass is only submitted after the host has seen the row of the one before it)
    const bool via_aql = h->aql_on && later_pass && host_fanin && !from_plane;
    if (via_aql) { h->aql_passes++; h->cnt.track_passes_own_queue++; }
    HIP_TRY(h, launch_track_folded(h->stream, p, h->dw, h->samples, h->partials, h->fold_ctr + (seq & 1ull) * track_fold_counter_words(), h->red_dev,
                                   use_rccl ? nullptr : host_row, host_fanin ? h->shard_host : nullptr, dev_word, seq,
                                   use_peer ? &px : nullptr, h->track_stamps, via_aql ? &h->aql : nullptr));
    if (from_plane && h->deferred_list_samples) h->deferred.samples_listed = true;
    const auto tp2 = h->track_profile ? std::chrono::steady_clock::now() : tp0;
    if (h->track_profile) {
        h->tp_fill += std::chrono::duration<double, std::nano>(tp1 - tp0).count();
        h->tp_launch += std::chrono::duration<double, std::nano>(tp2 - tp1).count();
        h->tp_passes++;
    }
    if (h->timing_track) HIP_TRY(h, hipEventRecord(h->ev_track.b, h->stream));
    if (use_rccl) {
        std::string cerr;
        if (!h->comm.allreduce_sum_f64(h->red_dev, kRedAllreduce, h->stream, &cerr))
            return fail(h, TSDF_E_COMM, "RCCL all-reduce failed: %s", cerr.c_str());
        if (h->poll) HIP_TRY(h, launch_track_publish(h->stream, h->red_dev, h->red_host, seq));
        else HIP_TRY(h, hipMemcpyAsync(h->red_host, h->red_dev, kRedWidth * sizeof(double), hipMemcpyDeviceToHost,
                                       h->stream));
    }
    bool arrived = false;
    if (host_fanin) {
        const int ns = track_num_shards(h->n_samples);
        const auto t0 = std::chrono::steady_clock::now();
        // a shard's slot: 40 {value, word} pairs, each written by ONE 16-byte store (track_kernel).  A value is good when
        // the word next to it is this pass's: read the word, then the value (loads stay in order on the host).
        struct Pair { double v; unsigned long long w; };
        const volatile Pair* pairs = reinterpret_cast<const volatile Pair*>(h->shard_host);
        double rows[kTrackShards][kPartWidth];
        bool all = true;
        for (int sh = 0; sh < ns && all; ++sh) {
            const volatile Pair* sp = pairs + (size_t)sh * (kShardSlotDoubles / 2);
            for (int e = kPartWidth - 1; e >= 0 && all; --e) {
                // (value, word) may be read in two pieces here and -- not architecturally excluded -- written in two pieces
                // on the way: whichever half is stale, the pair does not validate and is read again
                for (unsigned spins = 0;; ++spins) {
                    const unsigned long long w = __atomic_load_n(&sp[e].w, __ATOMIC_ACQUIRE);
                    const unsigned long long vb = __atomic_load_n(reinterpret_cast<const volatile unsigned long long*>(&sp[e].v), __ATOMIC_ACQUIRE);
                    if (w == shard_pair_word(vb, seq)) { std::memcpy(&rows[sh][e], &vb, sizeof vb); break; }
                    if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) { all = false; break; }
                }
            }
        }
        if (!all) {                                            // a shard row did not show up in time: synchronise for real
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            h->aql.wait_idle();
            for (int sh = 0; sh < ns; ++sh)
                for (int e = 0; e < kPartWidth; ++e) {
                    const volatile Pair* sp = pairs + (size_t)sh * (kShardSlotDoubles / 2);
                    unsigned long long vb;
                    { const double v = sp[e].v; std::memcpy(&vb, &v, sizeof vb); }
                    if (sp[e].w != shard_pair_word(vb, seq)) return fail(h, TSDF_E_HIP, "tracker fan-in: shard %d of pass %llu never reached the host", sh, seq);
                    std::memcpy(&rows[sh][e], &vb, sizeof vb);
                }
        }
        if (h->track_profile) h->tp_wait += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tp2).count();
        // shard order, as the device's last workgroup adds them: the same bits in every exchange mode
        double tot[kPartWidth];
        bool stale = false;
        for (int e = 0; e < kPartWidth; ++e) {
            double v = rows[0][e];
            for (int sh = 1; sh < ns; ++sh) v += rows[sh][e];
            tot[e] = v;
        }
        for (int sh = 0; sh < ns; ++sh) stale |= rows[sh][kPartWidth - 1] != (double)(seq & 0xFFFFFFFFFFFFull);
        track_unpack_row(tot, h->red_host);
        if (stale) h->red_host[27] = std::nan("");
        arrived = true;
    } else if (dev_publish_shm) {
        int rc2 = shm_fan_in(h, seq, kRedAllreduce);
        if (rc2) return rc2;
        arrived = true;
    } else if (h->poll) {
        // The last workgroup (or the publish kernel) releases the pass number after the row (system scope); spinning on
        // it saves the runtime's completion-signal path.  Bounded: fall back to a real synchronisation.
        volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(h->red_host + kRedWidth);
        const auto t0 = std::chrono::steady_clock::now();
        const auto limit = std::chrono::milliseconds(use_peer ? kPeerTimeoutMs + 1000 : use_rccl ? 2000 : 5);
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > limit) break;
        }
    }
    if (!arrived) HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (use_shm && !dev_publish_shm) {
        // this rank's finished row is in red_host: publish it with a host store, then add everybody's rows
        char* slot = h->shm.base + shm_slot_offset(h, h->shm.rank, seq);
        std::memcpy(slot, h->red_host, kRedWidth * sizeof(double));
        __atomic_store_n(reinterpret_cast<unsigned long long*>(slot + kRedWidth * sizeof(double)), shm_word(h, seq), __ATOMIC_RELEASE);
        int rc2 = shm_fan_in(h, seq, kRedAllreduce);
        if (rc2) return rc2;
    }
    if (h->red_host[27] != h->red_host[27]) {
        unsigned long long bits;
        std::memcpy(&bits, &h->red_host[27], sizeof bits);
        if (use_peer && bits == kRowPoisonPeerTimeout)
            return fail(h, TSDF_E_COMM, "peer exchange: not every rank delivered its row of pass %llu within %d ms", seq, kPeerTimeoutMs);
        return fail(h, TSDF_E_HIP, "tracker fan-in: a partial row stayed stale through two cache invalidations (hand-off protocol violated)%s",
                    use_peer ? " on one of the ranks" : "");
    }
    if (h->timing_track) {
        float ms = 0.f;
        hipError_t te = hipEventElapsedTime(&ms, h->ev_track.a, h->ev_track.b);
        if (te == hipErrorNotReady) {
            HIP_TRY(h, hipEventSynchronize(h->ev_track.b));
            te = hipEventElapsedTime(&ms, h->ev_track.a, h->ev_track.b);
        }
        HIP_TRY(h, te);
        h->tm.track_ms += ms;
        h->tm.track_launches++;
    }
    h->cnt.track_iterations++;
    h->cnt.track_in_grid += (int64_t)h->red_host[30];
    h->cnt.track_terms += (int64_t)h->red_host[27];
    if (reduce_ranks && !use_rccl && !use_shm && h->hook) {
        if (h->hook(h->red_host, kRedAllreduce, h->hook_ctx) != 0)
            return fail(h, TSDF_E_COMM, "all-reduce hook reported failure");
    }
    if (h->red_host[28] > 0.0)
        return fail(h, TSDF_E_HALO,
                    "%.0f tracking look-ups left the stored layers [%d,%d) of this rank: halo=%d is too small",
                    h->red_host[28], h->grid.xs, h->grid.xe, h->cfg.halo);
    return TSDF_OK;
}

// Cumulative device counters into h->counters_host (synchronises the main stream): the item count comes from the
// counter block, the updated-voxel counts are the sums of the integrate workgroups' own words.
int fetch_counters(tsdf_handle* h) {
    const size_t nw = 2 * (size_t)h->integrate_blocks;
    HIP_TRY(h, hipMemcpyAsync(h->counters_host, h->counters, kNumCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->wg_counts_host, h->wg_counts, 13 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->integrate_debug & 256) {       // stage profile of debug builds: shader-clock cycles per pipeline step, first wavefront of every workgroup
        if (h->integrate_debug & 2048) {     // per workgroup: steps and loop time (10 ns ticks) of its first wavefront, cumulative
            for (size_t b = 0; b < (size_t)h->integrate_blocks; ++b)
                std::fprintf(stderr, "WGLOOP %zu %llu %llu\n", b, h->wg_counts_host[nw + 6 * (4 * b) + 3], h->wg_counts_host[nw + 6 * (4 * b) + 5]);
        }
        for (int wv = 0; wv < 4; ++wv) {
            unsigned long long t[6] = {0, 0, 0, 0, 0, 0};
            for (size_t b = 0; b < (size_t)h->integrate_blocks; ++b)
                for (int q = 0; q < 6; ++q) t[q] += h->wg_counts_host[nw + 6 * (4 * b + wv) + q];
            if (t[3])
                std::fprintf(stderr, "STAGES wave %d steps %llu cycles_per_step S1 %.1f S2 %.1f S3 %.1f loop %.1f (%.3f us per step, %.0f MHz)\n", wv, t[3],
                             (double)t[0] / t[3], (double)t[1] / t[3], (double)t[2] / t[3], (double)t[4] / t[3], 0.01 * (double)t[5] / t[3],
                             t[5] ? (double)t[4] / (0.01 * (double)t[5]) : 0.0);
        }
    }
    {   // TSDF_LIVE_HIST=1 with a -DTSDF_LIVE_HISTOGRAM build: work items by the number of lanes they update (cumulative)
        static const bool live_hist = [] { const char* e = std::getenv("TSDF_LIVE_HIST"); return e && std::atoi(e) != 0; }();
        if (live_hist) {
            unsigned long long t[6] = {0, 0, 0, 0, 0, 0};
            for (size_t w = 0; w < 4 * (size_t)h->integrate_blocks; ++w)
                for (int q = 0; q < 6; ++q) t[q] += h->wg_counts_host[nw + 6 * w + q];
            std::fprintf(stderr, "LIVEHIST items by updated lanes: 0: %llu  1-16: %llu  17-32: %llu  33-48: %llu  49-63: %llu  64: %llu\n", t[0], t[1], t[2], t[3], t[4], t[5]);
        }
    }
    {   // TSDF_WG_FINISH=1 with a -DTSDF_WG_FINISH build: when the wavefronts of the LAST integrate launch ended their item loops
        static const bool wg_finish = [] { const char* e = std::getenv("TSDF_WG_FINISH"); return e && std::atoi(e) != 0; }();
        if (wg_finish) {
            const size_t nwv = 4 * (size_t)h->integrate_blocks;
            unsigned long long t0 = ~0ull;
            for (size_t w = 0; w < nwv; ++w) if (h->wg_counts_host[nw + 6 * w]) t0 = std::min(t0, h->wg_counts_host[nw + 6 * w]);
            std::vector<double> end, start;
            double by_gen[8] = {0, 0, 0, 0, 0, 0, 0, 0}; size_t n_gen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t w = 0; w < nwv; ++w) {
                if (!h->wg_counts_host[nw + 6 * w]) continue;
                const double e = 0.01 * (double)(h->wg_counts_host[nw + 6 * w + 1] - t0);
                end.push_back(e); start.push_back(0.01 * (double)(h->wg_counts_host[nw + 6 * w] - t0));
                const size_t gen = std::min<size_t>(7, ((w / 4) >> 3) / 32);
                by_gen[gen] += e; n_gen[gen] += 1;
            }
            std::sort(end.begin(), end.end()); std::sort(start.begin(), start.end());
            if (!end.empty()) {
                double mean = 0; for (double e : end) mean += e; mean /= (double)end.size();
                auto q = [&](const std::vector<double>& v, double f) { return v[std::min(v.size() - 1, (size_t)(f * (double)v.size()))]; };
                std::fprintf(stderr, "WGFINISH wavefronts %zu: loop start p50 %.1f max %.1f us; loop end mean %.1f min %.1f p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f us;",
                             end.size(), q(start, 0.5), start.back(), mean, end.front(), q(end, 0.1), q(end, 0.25), q(end, 0.5), q(end, 0.75), q(end, 0.9), end.back());
                std::fprintf(stderr, " by workgroup generation (index in the XCD / 32):");
                for (int g = 0; g < 8; ++g) if (n_gen[g]) std::fprintf(stderr, " %.1f", by_gen[g] / (double)n_gen[g]);
                double x_max[8] = {0, 0, 0, 0, 0, 0, 0, 0}, x_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; size_t x_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t w = 0; w < nwv; ++w) {
                    if (!h->wg_counts_host[nw + 6 * w]) continue;
                    const double e = 0.01 * (double)(h->wg_counts_host[nw + 6 * w + 1] - t0);
                    const size_t x = (w / 4) & 7;
                    x_max[x] = std::max(x_max[x], e); x_sum[x] += e; x_n[x] += 1;
                }
                std::fprintf(stderr, "; by XCD mean/max:");
                for (int x = 0; x < 8; ++x) if (x_n[x]) std::fprintf(stderr, " %.1f/%.1f", x_sum[x] / (double)x_n[x], x_max[x]);
                std::fprintf(stderr, "\n");
            }
        }
    }
    unsigned long long own = 0, halo = 0;
    for (size_t b = 0; b < nw; b += 2) { own += h->wg_counts_host[b]; halo += h->wg_counts_host[b + 1]; }
    {   // TSDF_LIST_STATS=1: how much of the work list missed its band's predicted region (cumulative)
        static const bool list_stats = [] { const char* e = std::getenv("TSDF_LIST_STATS"); return e && std::atoi(e) != 0; }();
        if (list_stats)
            std::fprintf(stderr, "LISTSTATS items %llu in the overflow region %llu\n", h->counters_host[kCntItems], h->counters_host[kCntOverflowItems]);
    }
    if (h->integrate_debug & 4096) {      // load-balance experiment of debug builds: per workgroup {updated voxels, 10 ns ticks}, cumulative
        for (size_t b = 0; b < nw; b += 2)
            std::fprintf(stderr, "WGT %zu %llu %llu\n", b / 2, h->wg_counts_host[b + 1], h->wg_counts_host[b]);
    }
    h->counters_host[kCntUpdatedOwned] = own;
    h->counters_host[kCntUpdatedHalo] = halo;
    return TSDF_OK;
}

void unpack_normal_equations(const double* row, double A[36], double b[6]) {
    int e = 0;
    for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) { A[6 * a + c] = row[e]; A[6 * c + a] = row[e]; ++e; }
    for (int a = 0; a < 6; ++a) b[a] = row[21 + a];
}

int check_ready(tsdf_handle* h, bool need_frame) {
    if (!h) return TSDF_E_BADARG;
    if (need_frame && !h->have_frame) return fail(h, TSDF_E_NO_FRAME, "no frame: call tsdf_set_frame first");
    return bind_device(h);
}

}  // namespace

// =================================================================================================
extern "C" {

int tsdf_abi_version(void) { return TSDF_ABI_VERSION; }

void tsdf_default_config(tsdf_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->m = 256; c->width = 6.0f; c->height = 6.0f; c->depth = 3.5f;      // sdf_reconstruction.cpp:83-85
    c->origin[0] = -3.0; c->origin[1] = -3.0; c->origin[2] = -0.5;
    c->delta = 0.3f; c->epsilon = 0.025f;
    c->gn_max_iter = 20; c->max_twist_diff = 0.001f; c->v_h = 1.0f; c->w_h = 0.01f;   // :88
    c->pixel_stride = 3;                                                   // camera_tracking.cpp:162-163
    c->stale_carry = 1;
    c->carry_threads = 1;
    c->with_color = 1;
    c->slab_x0 = 0; c->slab_x1 = 0; c->halo = 0; c->device = 0;
}

const char* tsdf_strerror(int s) {
    switch (s) {
        case TSDF_OK: return "ok";
        case TSDF_E_BADARG: return "bad argument";
        case TSDF_E_NO_DEVICE: return "no usable HIP device";
        case TSDF_E_HIP: return "HIP runtime error";
        case TSDF_E_NO_INTRINSICS: return "camera intrinsics not set";
        case TSDF_E_NO_FRAME: return "no frame set";
        case TSDF_E_SINGULAR: return "normal equations singular or pose not finite";
        case TSDF_E_NO_SAMPLES: return "no valid tracking samples";
        case TSDF_E_HALO: return "slab halo too small";
        case TSDF_E_COMM: return "all-reduce failure";
        case TSDF_E_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char* tsdf_last_error(const tsdf_handle* h) {
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_create_error.c_str();
}

int tsdf_slab_range(int32_t m, int32_t nranks, int32_t rank, int32_t* x0, int32_t* x1) {
    if (m <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || !x0 || !x1) return TSDF_E_BADARG;
    const int64_t q = m / nranks, r = m % nranks;      // the first r ranks get one extra layer
    const int64_t lo = q * rank + (rank < r ? rank : r);
    *x0 = (int32_t)lo;
    *x1 = (int32_t)(lo + q + (rank < r ? 1 : 0));
    return TSDF_OK;
}

// Slabs of equal WORK instead of equal thickness.  The cost of a rank is the weight of the layers it STORES (slab + halo
// per side: halo layers are integrated too); boundaries minimise the largest cost.  Monotone greedy under a bisected
// bound: every rank computes the same boundaries from the same weights.
int tsdf_slab_range_weighted(int32_t m, int32_t nranks, int32_t rank, int32_t halo, const double* w, int32_t* x0, int32_t* x1) {
    if (m <= 0 || nranks <= 0 || nranks > m || rank < 0 || rank >= nranks || halo < 0 || !w || !x0 || !x1) return TSDF_E_BADARG;
    try {                                    // (the vectors below: nothing may throw across the C ABI)
    std::vector<double> pre((size_t)m + 1, 0.0);
    for (int32_t i = 0; i < m; ++i) {
        if (!(w[i] >= 0.0) || !std::isfinite(w[i])) return TSDF_E_BADARG;
        pre[(size_t)i + 1] = pre[(size_t)i] + w[i];
    }
    if (!(pre[(size_t)m] > 0.0)) return tsdf_slab_range(m, nranks, rank, x0, x1);      // no information: equal thickness
    auto cost = [&](int32_t a, int32_t b) {                       // stored layers of the slab [a, b)
        const int32_t lo = a - halo < 0 ? 0 : a - halo, hi = b + halo > m ? m : b + halo;
        return pre[(size_t)hi] - pre[(size_t)lo];
    };
    std::vector<int32_t> cut((size_t)nranks + 1, 0);
    auto greedy = [&](double T, std::vector<int32_t>& c) {
        int32_t x = 0;
        c[0] = 0;
        for (int32_t r = 0; r < nranks; ++r) {
            const int32_t last = m - (nranks - 1 - r);            // leave a layer for every rank behind this one
            if (r == nranks - 1) { if (cost(x, m) > T) return false; c[(size_t)r + 1] = m; return true; }
            if (cost(x, x + 1) > T) return false;
            int32_t lo = x + 1, hi = last;                        // the largest b in [x+1, last] with cost(x, b) <= T
            while (lo < hi) { const int32_t mid = lo + (hi - lo + 1) / 2; if (cost(x, mid) <= T) lo = mid; else hi = mid - 1; }
            x = lo;
            c[(size_t)r + 1] = x;
        }
        return true;
    };
    double lo = 0.0, hi = pre[(size_t)m];
    for (int it = 0; it < 100 && hi - lo > 1e-12 * pre[(size_t)m]; ++it) {
        const double mid = 0.5 * (lo + hi);
        std::vector<int32_t> c((size_t)nranks + 1, 0);
        if (greedy(mid, c)) hi = mid; else lo = mid;
    }
    if (!greedy(hi, cut)) return TSDF_E_BADARG;
    // the greedy cut loads the early ranks to the bound and leaves the last ones light: pull the cuts back while no cost
    // exceeds the bound, so that thickness is shared where work is not (rank memory stays reasonable)
    for (int32_t r = nranks - 1; r >= 1; --r) {
        int32_t lo_c = cut[(size_t)r - 1] + 1, hi_c = cut[(size_t)r];          // the smallest cut[r] that keeps rank r within the bound
        while (lo_c < hi_c) { const int32_t mid = lo_c + (hi_c - lo_c) / 2; if (cost(mid, cut[(size_t)r + 1]) <= hi) hi_c = mid; else lo_c = mid + 1; }
        // half-way between "as early as the bound allows" and the greedy position
        cut[(size_t)r] = lo_c + (cut[(size_t)r] - lo_c) / 2;
    }
    *x0 = cut[(size_t)rank]; *x1 = cut[(size_t)rank + 1];
    return TSDF_OK;
    } catch (...) {
        return TSDF_E_NOMEM;
    }
}

// Expected integration work per x layer for one camera pose: the voxels of the layer inside the view frustum up to
// max_depth (rows sampled every `step`-th j, their k intervals from the same affine tests list_rows_kernel uses), in units of
// 64-voxel work items, plus a floor for the per-row work every stored layer costs.  ADDS to weights[0..m): poses accumulate.
int tsdf_frustum_layer_weights(const tsdf_config* c, const double K[9], int32_t width, int32_t height, const double rot[9],
                               const double trans[3], float max_depth, double* weights) {
    if (!c || !K || !rot || !trans || !weights || c->m <= 0 || width <= 0 || height <= 0 || !(max_depth > 0.f)) return TSDF_E_BADARG;
    hm::Pose ps;
    hm::set_pose(ps, rot, trans);
    const int m = c->m;
    const double cw = (double)(c->width / (float)m), ch = (double)(c->height / (float)m), cd = (double)(c->depth / (float)m);
    const int step = m >= 256 ? m / 128 : 1;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < m; ++i) {
        const double gx = cw * (i + 0.5) + c->origin[0];
        double voxels = 0.0;
        for (int j = step / 2; j < m; j += step) {
            const double gy = ch * (j + 0.5) + c->origin[1], gz0 = cd * 0.5 + c->origin[2];
            double Q0[3], Q1[3];
            for (int a = 0; a < 3; ++a) {
                Q0[a] = ps.rot_inv[3 * a] * gx + ps.rot_inv[3 * a + 1] * gy + ps.rot_inv[3 * a + 2] * gz0 + ps.rot_inv_trans[a];
                Q1[a] = ps.rot_inv[3 * a + 2] * cd;
            }
            double lo = 0.0, hi = (double)m;
            auto clip = [&](double a, double b) {                 // a + k b >= 0
                if (b > 0.0) { const double t = -a / b; if (t > lo) lo = t; }
                else if (b < 0.0) { const double t = -a / b; if (t < hi) hi = t; }
                else if (a < 0.0) { lo = 1.0; hi = 0.0; }
            };
            clip(Q0[2] - 0.05, Q1[2]);                                                    // in front of the camera
            clip((double)max_depth - Q0[2], -Q1[2]);                                      // within the sensor's range
            clip(fx * Q0[0] + (cx + 0.5) * Q0[2], fx * Q1[0] + (cx + 0.5) * Q1[2]);       // u >= -0.5
            clip(-(fx * Q0[0] + (cx - (width - 0.5)) * Q0[2]), -(fx * Q1[0] + (cx - (width - 0.5)) * Q1[2]));
            clip(fy * Q0[1] + (cy + 0.5) * Q0[2], fy * Q1[1] + (cy + 0.5) * Q1[2]);
            clip(-(fy * Q0[1] + (cy - (height - 0.5)) * Q0[2]), -(fy * Q1[1] + (cy - (height - 0.5)) * Q1[2]));
            if (hi > lo) voxels += (hi - lo) * step;
        }
        weights[i] += voxels / 64.0 + 0.02 * (double)m * (double)m / 64.0;
    }
    return TSDF_OK;
}

int32_t tsdf_halo_for(const tsdf_config* c, float max_range) {
    if (!c || c->m <= 0 || !(c->width > 0)) return -1;
    const double per_m = (double)c->m / (double)c->width;
    return (int32_t)std::ceil((double)c->w_h * (double)max_range * per_m) + (int32_t)std::ceil((double)c->v_h) + 2;
}

// camera_tracking.cpp:11-17: the finite-difference denominators are float quotients formed once from v_h / w_h; they
// must follow the steps whenever those change (tsdf_create, tsdf_set_tracker_params), or the kernel perturbs by the new
// step and divides by the old one.
static void set_step_denominators(tsdf_handle* h, float v_h, float w_h) {
    const Grid& g = h->grid;
    const float v_h2 = 2 * v_h;
    h->v_h2_w = v_h2 / g.m_div_w;
    h->v_h2_h = v_h2 / g.m_div_h;
    h->v_h2_d = v_h2 / g.m_div_d;
    h->wh2 = 2 * w_h;
}

int tsdf_create(const tsdf_config* cfg, tsdf_handle** out) {
    if (!cfg || !out) return fail(nullptr, TSDF_E_BADARG, "tsdf_create: null argument");
    *out = nullptr;
    if (cfg->m < 2 || cfg->m > 4096 || !(cfg->width > 0) || !(cfg->height > 0) || !(cfg->depth > 0) ||
        cfg->pixel_stride < 1 || cfg->gn_max_iter < 0 || cfg->halo < 0)
        return fail(nullptr, TSDF_E_BADARG, "tsdf_create: bad config (m=%d stride=%d)", cfg->m, cfg->pixel_stride);
    int32_t x0 = cfg->slab_x0, x1 = cfg->slab_x1;
    if (x0 == 0 && x1 == 0) x1 = cfg->m;
    if (x0 < 0 || x1 > cfg->m || x0 >= x1) return fail(nullptr, TSDF_E_BADARG, "tsdf_create: bad slab [%d,%d)", x0, x1);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, TSDF_E_NO_DEVICE, "no HIP device visible (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, TSDF_E_NO_DEVICE, "device ordinal %d out of range (%d devices)", cfg->device, ndev);

    tsdf_handle* h = new (std::nothrow) tsdf_handle();
    if (!h) return fail(nullptr, TSDF_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->cfg.slab_x0 = x0; h->cfg.slab_x1 = x1;
    h->device = cfg->device;

    Grid& g = h->grid;
    g.m = cfg->m;
    g.own_x0 = x0; g.own_x1 = x1;
    g.xs = x0 - cfg->halo < 0 ? 0 : x0 - cfg->halo;
    g.xe = x1 + cfg->halo > cfg->m ? cfg->m : x1 + cfg->halo;
    g.cell_w = cfg->width / ((float)cfg->m);       // sdf.h:154-156
    g.cell_h = cfg->height / ((float)cfg->m);
    g.cell_d = cfg->depth / ((float)cfg->m);
    g.m_div_w = cfg->m / cfg->width;               // sdf.cpp:19-21
    g.m_div_h = cfg->m / cfg->height;
    g.m_div_d = cfg->m / cfg->depth;
    std::memcpy(g.origin, cfg->origin, sizeof g.origin);
    g.delta = cfg->delta; g.epsilon = cfg->epsilon;

    // camera_tracking.cpp:5-17
    const double rot0[9] = {1, 0, 0, 0, 0, -1, 0, -1, 0};
    const double trans0[3] = {0, 0, 1};
    hm::set_pose(h->pose, rot0, trans0);
    set_step_denominators(h, cfg->v_h, cfg->w_h);

    auto bail = [&](int code) { std::string msg = h->err; tsdf_destroy(h); fail(nullptr, code, "%s", msg.c_str()); return code; };
#define CREATE_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t e2__ = (expr);                                                                     \
        if (e2__ != hipSuccess) {                                                                     \
            fail(h, TSDF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e2__));                     \
            return bail(e2__ == hipErrorOutOfMemory ? TSDF_E_NOMEM : TSDF_E_HIP);                     \
        }                                                                                             \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&h->fstream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_frame, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_samples, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_queued, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_stage_done[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_stage_done[1], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_buf_used[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_buf_used[1], hipEventDisableTiming));
    h->n_stored = (int64_t)(g.xe - g.xs) * g.m * g.m;
    // Padding voxels {D = 0, W = 0} around the volume (16 in front: keeps the 128-byte alignment of the rows; 2 behind).
    // Tracker look-ups read the corner pair (k, k+1) with one 16-byte load at k in [-1, m-1]: at the two ends of a row
    // that touches the neighbouring row or, for the first / last row, this padding; the pair behind the volume also
    // stands in for rows that are not stored (W = 0 makes the reference's own W > 0 test skip it).
    CREATE_TRY(hipMalloc((void**)&h->dw_alloc, ((size_t)h->n_stored + kVolumePadFront + 2) * sizeof(float2)));
    CREATE_TRY(hipMemset(h->dw_alloc, 0, kVolumePadFront * sizeof(float2)));
    h->dw = h->dw_alloc + kVolumePadFront;
    CREATE_TRY(hipMemset(h->dw + h->n_stored, 0, 2 * sizeof(float2)));
    if (cfg->with_color) CREATE_TRY(hipMalloc((void**)&h->crgb, (size_t)h->n_stored * sizeof(float4)));
    CREATE_TRY(hipMalloc((void**)&h->counters, kNumCounters * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(h->counters, 0, kNumCounters * sizeof(unsigned long long), h->stream));
    CREATE_TRY(hipHostMalloc((void**)&h->counters_host, kNumCounters * sizeof(unsigned long long), hipHostMallocDefault));
    CREATE_TRY(hipMalloc((void**)&h->worklist, integrate_worklist_bytes(g)));
    CREATE_TRY(hipMemsetAsync(h->worklist, 0, integrate_worklist_bytes(g), h->stream));
    CREATE_TRY(hipMalloc((void**)&h->work_count, integrate_bookkeeping_words() * sizeof(unsigned)));
    CREATE_TRY(hipMemsetAsync(h->work_count, 0, integrate_bookkeeping_words() * sizeof(unsigned), h->stream));
    {
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, h->device));
        { const char* dbg = std::getenv("TSDF_DEBUG_INTEGRATE"); h->integrate_debug = dbg ? std::atoi(dbg) : 0; }
        // TSDF_INTEGRATE_KERNEL=queue selects round 4's dense-batch kernel (integrate_queue_kernel: bit-identical volume,
        // 25 % fewer vector instructions, but 10 % SLOWER on MI355X -- DESIGN.md section 7; kept for same-box comparisons);
        // the default is the item-at-a-time integrate_kernel
        const char* kk = std::getenv("TSDF_INTEGRATE_KERNEL");
        h->integrate_queue = integrate_queue_fits(g) && kk && std::strcmp(kk, "queue") == 0;
        const char* env = std::getenv("TSDF_INTEGRATE_BLOCKS_PER_CU");
        const int per_cu = env ? std::atoi(env) : integrate_blocks_per_cu(h->integrate_queue);
        h->integrate_blocks = (prop.multiProcessorCount * (per_cu > 0 ? per_cu : 4) + 7) / 8 * 8;   // whole XCD groups
        h->integrate_cus = prop.multiProcessorCount;
        { const char* ev = std::getenv("TSDF_INTEGRATE_GRID_BY_WORK"); h->integrate_grid_by_work = !(ev && std::atoi(ev) == 0); }
        // the measurement builds index their per-wavefront words by the full grid
        if (h->integrate_debug || std::getenv("TSDF_WG_FINISH") || std::getenv("TSDF_LIVE_HIST")) h->integrate_grid_by_work = false;
    }
    // 2 words per workgroup (updated voxels: owned, halo) + 4 x 6 more behind them for the stage profile of debug builds
    CREATE_TRY(hipMalloc((void**)&h->wg_counts, 26 * (size_t)h->integrate_blocks * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(h->wg_counts, 0, 26 * (size_t)h->integrate_blocks * sizeof(unsigned long long), h->stream));
    CREATE_TRY(hipHostMalloc((void**)&h->wg_counts_host, 26 * (size_t)h->integrate_blocks * sizeof(unsigned long long), hipHostMallocDefault));
    CREATE_TRY(hipMalloc((void**)&h->red_dev, kRedWidth * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&h->red_host, (kRedWidth + 2) * sizeof(double), hipHostMallocDefault));
    std::memset(h->red_host, 0, (kRedWidth + 2) * sizeof(double));
    CREATE_TRY(hipHostMalloc((void**)&h->release_host, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    h->release_host[0] = h->release_host[1] = 0ull;
    h->release_host[2] = ~0ull;              // work items of the last integrate launch (none yet: the full grid)
    { const char* ev = std::getenv("TSDF_NO_POLL"); h->poll = !(ev && std::atoi(ev) != 0); }
    { const char* ev = std::getenv("TSDF_DEFER_PACK"); h->defer_device_pack = !(ev && std::atoi(ev) == 0); h->deferred_list_samples = !(ev && std::atoi(ev) == 2); }
    { const char* ev = std::getenv("TSDF_HOST_FOLD"); h->host_fold = !(ev && std::atoi(ev) == 0); }
    CREATE_TRY(hipHostMalloc((void**)&h->shard_host, (size_t)kTrackShards * kShardSlotDoubles * sizeof(double), hipHostMallocDefault));
    std::memset(h->shard_host, 0, (size_t)kTrackShards * kShardSlotDoubles * sizeof(double));
    { const char* ev = std::getenv("TSDF_HOST_FANIN"); h->host_fanin = !(ev && std::atoi(ev) == 0); }
    { const char* ev = std::getenv("TSDF_TRACK_STAMPS");      // diagnosis: where inside the launch does a tracker pass spend its time?
      if (ev && std::atoi(ev) != 0) {
          CREATE_TRY(hipMalloc((void**)&h->track_stamps, 8 * (size_t)kTrackStampBlocks * sizeof(unsigned long long)));
          CREATE_TRY(hipMemsetAsync(h->track_stamps, 0, 8 * (size_t)kTrackStampBlocks * sizeof(unsigned long long), h->stream));
      } }
    { const char* ev = std::getenv("TSDF_TRACK_PROFILE"); h->track_profile = ev && std::atoi(ev) != 0; }
    { const char* ev = std::getenv("TSDF_STAGE_PROFILE"); h->sp.on = ev && std::atoi(ev) != 0; }
    CREATE_TRY(hipMalloc((void**)&h->fold_ctr, 2 * track_fold_counter_words() * sizeof(unsigned)));
    CREATE_TRY(hipMemsetAsync(h->fold_ctr, 0, 2 * track_fold_counter_words() * sizeof(unsigned), h->stream));
    {
        const char* ev = std::getenv("TSDF_AQL");
        if (!(ev && std::atoi(ev) == 0) && !h->track_stamps) {
            // the code object sits next to this library
            Dl_info info;
            std::string path;
            if (dladdr(reinterpret_cast<const void*>(&tsdf_abi_version), &info) && info.dli_fname) {
                path = info.dli_fname;
                const size_t slash = path.find_last_of('/');
                std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
                const std::string dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
                const size_t dot = base.rfind(".so");
                base = base == "libtsdf_hip.so" ? std::string("tsdf_kernels.hsaco") : base.substr(0, dot) + ".hsaco";
                path = dir + "/" + base;
            }
            std::string why;
            h->aql_on = !path.empty() && h->aql.init(h->device, path.c_str(), track_kernel_symbol_prefix(), track_kernel_explicit_arg_bytes(), &why);
            if (!h->aql_on && ev && std::atoi(ev) == 2)          // TSDF_AQL=2: say why the queue is not in use
                std::fprintf(stderr, "[tsdf] AQL queue for the tracker passes not in use: %s\n", why.c_str());
        }
    }
    CREATE_TRY(hipEventCreate(&h->ev_track.a));
    CREATE_TRY(hipEventCreate(&h->ev_track.b));
    CREATE_TRY(launch_fill(h->stream, g, h->dw, h->crgb, cfg->width + cfg->height + cfg->depth));   // sdf.cpp:29
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    *out = h;
    return TSDF_OK;
}

void tsdf_destroy(tsdf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->track_stamps) {
        // phase stamps of the LAST tracker pass: per workgroup, microseconds after the earliest workgroup's start
        std::vector<unsigned long long> st(8 * (size_t)kTrackStampBlocks);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(st.data(), h->track_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            const int nb = track_num_blocks(h->n_samples) < kTrackStampBlocks ? track_num_blocks(h->n_samples) : kTrackStampBlocks;
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < nb; ++b) if (st[8 * b] && st[8 * b] < t0) t0 = st[8 * b];
            const char* names[8] = {"start", "window classified", "own sample", "look-ups done", "row written", "arrived", "shard row out",
                                    "shard word released"};
            for (int k = 0; k < 8; ++k) {
                double mn = 1e30, mx = 0, sum = 0; int n = 0;
                for (int b = 0; b < nb; ++b) {
                    if (!st[8 * b + k] || st[8 * b + k] < t0) continue;
                    const double v = 0.01 * (double)(st[8 * b + k] - t0);
                    mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; ++n;
                }
                if (n) std::fprintf(stderr, "TRACK_STAMPS %-18s workgroups %4d  min %7.2f  mean %7.2f  max %7.2f us\n", names[k], n, mn, sum / n, mx);
            }
        }
        (void)hipFree(h->track_stamps);
    }
    if (h->sp.on && h->sp.frames) {
        const double f = (double)h->sp.frames * 1e3;
        std::fprintf(stderr, "STAGE_PROFILE frames %lld  us per frame: staging call %.1f  slowest worker's filling %.1f  first chunk ready %.1f  "
                             "inside hipMemcpyAsync %.1f  wait for the planes (previous copies) %.1f  hand-off to the staging thread %.1f  "
                             "tsdf_next_frame waits %.1f  workers %d\n",
                     h->sp.frames, h->sp.total / f, h->sp.fill_max / f, h->sp.first_chunk / f, h->sp.upload_calls / f, h->sp.sync_before / f,
                     h->sp.handoff / f, h->sp.next_wait / f, h->pool ? h->pool->parts() - 1 : 0);
    }
    if (h->sp.on && h->sp.aos_frames) {
        const double f = (double)h->sp.aos_frames * 1e3;
        std::fprintf(stderr, "AOS_PROFILE frames %lld  us per frame: tsdf_track_aos: checks + buffers %.1f  wait for the staging set %.1f  prepare %.1f  gather samples %.1f  issue copy + start staging %.1f  "
                             "Gauss-Newton loop %.1f  wait for the staging %.1f | tsdf_integrate_aos: repack normals + issue copy %.1f  compare cloud %.1f  "
                             "pack launch + events %.1f  tsdf_integrate call %.1f\n",
                     h->sp.aos_frames, h->sp.a_prep1 / f, h->sp.a_prep2 / f, h->sp.a_prep / f, h->sp.a_gather / f, h->sp.a_issue / f, h->sp.a_loop / f, h->sp.a_wait / f,
                     h->sp.b_normals / f, h->sp.b_verify / f, h->sp.b_issue / f, h->sp.b_integrate / f);
    }
    if (h->track_profile && h->tp_passes)
        std::fprintf(stderr, "TRACKPROFILE passes %lld  ns per pass: parameters %.0f  launch call %.0f  wait for the row %.0f  fold+solve+pose %.0f\n",
                     h->tp_passes, h->tp_fill / h->tp_passes, h->tp_launch / h->tp_passes, h->tp_wait / h->tp_passes, h->tp_post / h->tp_passes);
    if (h->qthread.joinable()) {                           // the staging thread of the frame queue
        {
            std::unique_lock<std::mutex> g(h->qmu);
            h->qcv.wait(g, [&] { return !h->qbusy; });
            h->qstop = true;
        }
        h->qcv.notify_all();
        h->qthread.join();
    }
    if (h->fstream) (void)hipStreamSynchronize(h->fstream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->aql.destroy();                        // (waits for its last packet)
    h->comm.destroy();
    peer_close(h);
    shm_close(h);
    free_frame(h);
    free_preproc(h);
    for (int b = 0; b < 2; ++b) {
        if (h->pn_buf[b]) (void)hipFree(h->pn_buf[b]);
        if (h->samples_buf[b]) (void)hipFree(h->samples_buf[b]);
        if (h->ev_buf_used[b]) (void)hipEventDestroy(h->ev_buf_used[b]);
    }
    if (h->ev_frame) (void)hipEventDestroy(h->ev_frame);
    if (h->ev_copied) (void)hipEventDestroy(h->ev_copied);
    if (h->ev_samples) (void)hipEventDestroy(h->ev_samples);
    if (h->ev_queued) (void)hipEventDestroy(h->ev_queued);
    for (int b = 0; b < 2; ++b) if (h->ev_stage_done[b]) (void)hipEventDestroy(h->ev_stage_done[b]);
    if (h->partials) (void)hipFree(h->partials);
    if (h->red_dev) (void)hipFree(h->red_dev);
    if (h->red_host) (void)hipHostFree(h->red_host);
    if (h->release_host) (void)hipHostFree(h->release_host);
    if (h->fold_ctr) (void)hipFree(h->fold_ctr);
    if (h->shard_host) (void)hipHostFree(h->shard_host);
    if (h->counters) (void)hipFree(h->counters);
    if (h->worklist) (void)hipFree(h->worklist);
    if (h->work_count) (void)hipFree(h->work_count);
    if (h->counters_host) (void)hipHostFree(h->counters_host);
    if (h->wg_counts) (void)hipFree(h->wg_counts);
    if (h->wg_counts_host) (void)hipHostFree(h->wg_counts_host);
    if (h->sample_vox) (void)hipFree(h->sample_vox);
    if (h->sample_val) (void)hipFree(h->sample_val);
    if (h->sample_ok) (void)hipFree(h->sample_ok);
    if (h->mesh_row_count) (void)hipFree(h->mesh_row_count);
    if (h->mesh_row_offset) (void)hipFree(h->mesh_row_offset);
    if (h->mesh_group_sum) (void)hipFree(h->mesh_group_sum);
    if (h->mesh_group_base) (void)hipFree(h->mesh_group_base);
    if (h->mesh_total) (void)hipHostFree(h->mesh_total);
    if (h->mesh_verts) (void)hipFree(h->mesh_verts);
    if (h->mesh_desc) (void)hipFree(h->mesh_desc);
    if (h->mesh_colors) (void)hipFree(h->mesh_colors);
    if (h->dw_alloc) (void)hipFree(h->dw_alloc);
    if (h->crgb) (void)hipFree(h->crgb);
    for (auto& ep : h->ev_pool) { if (ep.a) (void)hipEventDestroy(ep.a); if (ep.b) (void)hipEventDestroy(ep.b); }
    if (h->ev_track.a) (void)hipEventDestroy(h->ev_track.a);
    if (h->ev_track.b) (void)hipEventDestroy(h->ev_track.b);
    if (h->fstream) (void)hipStreamDestroy(h->fstream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int tsdf_get_config(const tsdf_handle* h, tsdf_config* cfg) {
    if (!h || !cfg) return TSDF_E_BADARG;
    *cfg = h->cfg;
    return TSDF_OK;
}

int tsdf_reset(tsdf_handle* h) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    HIP_TRY(h, launch_fill(h->stream, h->grid, h->dw, h->crgb, h->cfg.width + h->cfg.height + h->cfg.depth));
    const double rot0[9] = {1, 0, 0, 0, 0, -1, 0, -1, 0};
    const double trans0[3] = {0, 0, 1};
    hm::set_pose(h->pose, rot0, trans0);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TSDF_OK;
}

// ---- camera state ---------------------------------------------------------------------------------

int tsdf_set_intrinsics(tsdf_handle* h, const double K[9]) {
    if (!h || !K) return TSDF_E_BADARG;
    std::memcpy(h->K, K, sizeof h->K);
    h->have_K = true;
    return TSDF_OK;
}

int tsdf_set_camera_transformation(tsdf_handle* h, const double rot[9], const double trans[3]) {
    if (!h || !rot || !trans) return TSDF_E_BADARG;
    hm::set_pose(h->pose, rot, trans);
    return TSDF_OK;
}

int tsdf_set_tracker_params(tsdf_handle* h, int32_t gn_max_iter, float max_twist_diff, float v_h, float w_h) {
    if (!h) return TSDF_E_BADARG;
    if (gn_max_iter < 0 || !(v_h > 0.0f) || !(w_h > 0.0f) || !(max_twist_diff == max_twist_diff))
        return fail(h, TSDF_E_BADARG, "tsdf_set_tracker_params: bad argument (iterations %d, v_h %g, w_h %g)", gn_max_iter, (double)v_h, (double)w_h);
    h->cfg.gn_max_iter = gn_max_iter; h->cfg.max_twist_diff = max_twist_diff; h->cfg.v_h = v_h; h->cfg.w_h = w_h;
    set_step_denominators(h, v_h, w_h);
    return TSDF_OK;
}

int64_t tsdf_frame_serial(const tsdf_handle* h) { return h ? h->frame_serial : -1; }

int tsdf_get_pose(const tsdf_handle* h, double rot[9], double trans[3], double rot_inv[9], double rot_inv_trans[3]) {
    if (!h) return TSDF_E_BADARG;
    if (rot) std::memcpy(rot, h->pose.rot, sizeof h->pose.rot);
    if (trans) std::memcpy(trans, h->pose.trans, sizeof h->pose.trans);
    if (rot_inv) std::memcpy(rot_inv, h->pose.rot_inv, sizeof h->pose.rot_inv);
    if (rot_inv_trans) std::memcpy(rot_inv_trans, h->pose.rot_inv_trans, sizeof h->pose.rot_inv_trans);
    return TSDF_OK;
}

// ---- frames ------------------------------------------------------------------------------------------

namespace {
// Is this host pointer page-locked memory HIP can copy from directly (hipHostMalloc / hipHostRegister)?
// true when the whole range [p, p + bytes) is page-locked host memory: both ends are asked (a buffer of which only the
// first part lies in a hipHostRegister'ed range must go through the staging copy)
bool is_pinned_host(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    const void* ends[2] = {p, static_cast<const char*>(p) + (bytes - 1)};
    for (const void* q : ends) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// ---- array-of-structs clouds -> the pinned planes (tsdf_set_frame_aos / tsdf_queue_frame_aos) ------------------------
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

// cores this process may run on (the affinity mask: hardware_concurrency() reports the whole machine in a container)
inline int usable_cores() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return n; }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? (int)hc : 1;
}

HostPool* host_pool(tsdf_handle* h) {
    if (!h->pool) {
        // default: the usable cores less two (the caller's thread drives the GPU, one stays free), at most 12;
        // TSDF_HOST_THREADS overrides (1 = no workers)
        const int cores = usable_cores();
        int n = cores - 2 < 12 ? cores - 2 : 12;
        if (const char* e = std::getenv("TSDF_HOST_THREADS")) n = std::atoi(e);
        if (n > cores) n = cores;
        n = n < 1 ? 1 : n > 64 ? 64 : n;
        h->pool.reset(new (std::nothrow) HostPool(n - 1));
    }
    return h->pool.get();
}

// Pageable frame -> pinned staging -> HBM: the pool's workers fill the pinned planes (fill(i0, i1) writes pixels
// [i0, i1)), the calling thread issues the H2D copy.  Rounds 2-3 cut the frame into 4 chunks so that the DMA of chunk c
// ran while chunk c+1 was being filled; round 4 measured what that costs: a 640x480 frame's copies are bound by their
// NUMBER (~15-20 us each whatever the size), so 12 copies per frame lose more than the overlap wins (PCL clouds through
// the queue: 8 chunks 2490 frames/s, 4 chunks 3450, 2 chunks 4010, 1 chunk = 3 copies 4140).  Default now: one chunk,
// and the three planes in one block = ONE copy per frame.  TSDF_STAGE_CHUNKS keeps the pipelined form for large images.
hipError_t stage_and_upload(tsdf_handle* h, size_t npix, bool has_xyz, bool has_nrm, bool has_rgb,
                            const std::function<void(size_t, size_t)>& fill, int chunks_when_unset = 1) {
    constexpr int kMaxChunks = 16;
    // TSDF_STAGE_CHUNKS overrides; otherwise the caller's choice: 1 where only throughput counts (the frame queue), 2 where
    // the frame's LATENCY to the device is on the critical path (tsdf_track_frame_aos: medians 2570 / 2850 / 2790 / 2760
    // frames/s with 1 / 2 / 3 / 4 pieces, six alternations)
    static const int kEnvChunks = [] { const char* e = std::getenv("TSDF_STAGE_CHUNKS"); const int n = e ? std::atoi(e) : 0; return n < 0 ? 0 : n > kMaxChunks ? kMaxChunks : n; }();
    const int kChunks = kEnvChunks > 0 ? kEnvChunks : (chunks_when_unset < 1 ? 1 : chunks_when_unset > kMaxChunks ? kMaxChunks : chunks_when_unset);
    std::atomic<int> done[kMaxChunks];
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    hipError_t err = hipSuccess;
    using clk = std::chrono::steady_clock;
    const bool prof = h->sp.on;
    const clk::time_point t_begin = prof ? clk::now() : clk::time_point();
    std::atomic<long long> fill_ns_max{0};
    double upload_ns = 0, first_ns = 0;
    auto chunk_lo = [npix, kChunks](int c) { return npix * (size_t)c / (size_t)kChunks; };
    auto upload = [&](int c) {
        const size_t i0 = chunk_lo(c), n = chunk_lo(c + 1) - i0;
        if (!n || err != hipSuccess) return;
        const clk::time_point tu = prof ? clk::now() : clk::time_point();
        if (prof && c == 0) first_ns = std::chrono::duration<double, std::nano>(tu - t_begin).count();
        struct Tail { const bool on; const clk::time_point t0; double& acc; ~Tail() { if (on) acc += std::chrono::duration<double, std::nano>(clk::now() - t0).count(); } } tail{prof, tu, upload_ns};
        if (kChunks == 1 && has_xyz && has_nrm) {            // the whole frame: the planes are neighbours in both blocks -> one copy
            const size_t bytes = has_rgb ? frame_block_bytes(h->in_cap) - (h->in_cap - npix) * 3 : 2 * plane_stride_bytes(h->in_cap);
            err = hipMemcpyAsync(h->in_xyz, h->pin_xyz, bytes, hipMemcpyHostToDevice, h->fstream);
            return;
        }
        if (has_xyz) err = hipMemcpyAsync(h->in_xyz + 3 * i0, h->pin_xyz + 3 * i0, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream);
        if (has_nrm && err == hipSuccess) err = hipMemcpyAsync(h->in_nrm + 3 * i0, h->pin_nrm + 3 * i0, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream);
        if (has_rgb && err == hipSuccess) err = hipMemcpyAsync(h->in_rgb + 3 * i0, h->pin_rgb + 3 * i0, n * 3, hipMemcpyHostToDevice, h->fstream);
    };
    HostPool* const pool = host_pool(h);
    const std::function<void(int, int)> job = [&](int part, int parts) {
        if (parts == 1) {                                   // no workers: fill and issue in turn (the DMA still overlaps)
            for (int c = 0; c < kChunks; ++c) { fill(chunk_lo(c), chunk_lo(c + 1)); upload(c); }
        } else if (part == 0) {                             // the caller: HIP calls only
            for (int c = 0; c < kChunks; ++c) {
                while (done[c].load(std::memory_order_acquire) < parts - 1) std::this_thread::yield();
                upload(c);
            }
        } else {
            const size_t wk = (size_t)(part - 1), nw = (size_t)(parts - 1);
            long long mine = 0;
            for (int c = 0; c < kChunks; ++c) {
                const size_t c0 = chunk_lo(c), n = chunk_lo(c + 1) - c0;
                const clk::time_point tf = prof ? clk::now() : clk::time_point();
                fill(c0 + n * wk / nw, c0 + n * (wk + 1) / nw);
                if (prof) mine += std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tf).count();
                done[c].fetch_add(1, std::memory_order_release);
            }
            if (prof) { long long cur = fill_ns_max.load(); while (mine > cur && !fill_ns_max.compare_exchange_weak(cur, mine)) {} }
        }
    };
    if (pool) pool->run(job); else job(0, 1);
    if (prof) {
        h->sp.frames++;
        h->sp.total += std::chrono::duration<double, std::nano>(clk::now() - t_begin).count();
        h->sp.fill_max += (double)fill_ns_max.load();
        h->sp.first_chunk += first_ns;
        h->sp.upload_calls += upload_ns;
    }
    return err;
}
}  // namespace

int tsdf_set_frame(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t width, int32_t height) {
    if (!h || !xyz || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_set_frame: bad argument") : TSDF_E_BADARG;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_set_frame");
    int rc = bind_device(h);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    // Page-locked caller buffers are copied from directly (no staging pass through the library's own pinned buffers:
    // at 640x480 that memcpy is 8.3 MB per frame, longer than the frame's GPU work); the copies are complete when the
    // call returns, so the buffers are borrowed for the call only, as for pageable ones.
    const bool direct = is_pinned_host(xyz, npix * 12) && (!nrm || is_pinned_host(nrm, npix * 12)) && (!rgb || is_pinned_host(rgb, npix * 3));
    // the pinned staging buffers may still feed the previous frame's async copies (frame stream only: the
    // integration of the previous frame keeps running on the main stream meanwhile)
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    h->staged_xyz = false;                                 // until this frame's planes are complete on the device
    if (!direct) {
        static const bool samples_first = [] { const char* e = std::getenv("TSDF_SAMPLES_FIRST"); return !(e && std::atoi(e) == 0); }();
        if (samples_first) { rc = upload_samples_first(h, xyz, 12, 0, width); if (rc) return rc; }
        HIP_TRY(h, stage_and_upload(h, npix, true, nrm != nullptr, rgb != nullptr, [&](size_t i0, size_t i1) {
            std::memcpy(h->pin_xyz + 3 * i0, xyz + 3 * i0, (i1 - i0) * 3 * sizeof(float));
            if (nrm) std::memcpy(h->pin_nrm + 3 * i0, nrm + 3 * i0, (i1 - i0) * 3 * sizeof(float));
            if (rgb) std::memcpy(h->pin_rgb + 3 * i0, rgb + 3 * i0, (i1 - i0) * 3);
        }));
        h->staged_xyz = true;
        return run_pack(h, h->in_xyz, nrm ? h->in_nrm : nullptr, rgb ? h->in_rgb : nullptr, h->fstream, false, samples_first);
    }
    HIP_TRY(h, hipMemcpyAsync(h->in_xyz, xyz, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
    if (nrm) HIP_TRY(h, hipMemcpyAsync(h->in_nrm, nrm, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
    if (rgb) HIP_TRY(h, hipMemcpyAsync(h->in_rgb, rgb, npix * 3, hipMemcpyHostToDevice, h->fstream));
    h->staged_xyz = true;
    HIP_TRY(h, hipEventRecord(h->ev_copied, h->fstream));
    rc = run_pack(h, h->in_xyz, nrm ? h->in_nrm : nullptr, rgb ? h->in_rgb : nullptr, h->fstream);
    if (rc) return rc;
    HIP_TRY(h, hipEventSynchronize(h->ev_copied));      // the caller's buffers have been read
    return TSDF_OK;
}

// ---- two-deep frame queue ----------------------------------------------------------------------------------------
namespace {
void queue_thread_main(tsdf_handle* h) {
    (void)hipSetDevice(h->device);
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> g(h->qmu);
            h->qcv.wait(g, [&] { return h->qstop || h->qjob; });
            if (h->qstop) return;
            job.swap(h->qjob);
        }
        job();
        { std::lock_guard<std::mutex> g(h->qmu); h->qbusy = false; }
        h->qcv.notify_all();
    }
}

// the second set of pinned staging planes (the frame queue and tsdf_track_aos alternate between two sets)
int ensure_second_staging_set(tsdf_handle* h, size_t npix) {
    if (h->alt_cap >= npix) return TSDF_OK;
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    if (h->alt_xyz) (void)hipHostFree(h->alt_xyz);
    h->alt_xyz = h->alt_nrm = nullptr; h->alt_rgb = nullptr; h->alt_cap = 0;
    // the same block layout as the first set and the device block: sized like them (in_cap pixels)
    const size_t plane = plane_stride_bytes(h->in_cap);
    char* pin = nullptr;
    HIP_TRY(h, hipHostMalloc((void**)&pin, frame_block_bytes(h->in_cap), hipHostMallocDefault));
    h->alt_xyz = reinterpret_cast<float*>(pin); h->alt_nrm = reinterpret_cast<float*>(pin + plane); h->alt_rgb = reinterpret_cast<uint8_t*>(pin + 2 * plane);
    h->alt_cap = h->in_cap;
    h->stage_recorded[0] = h->stage_recorded[1] = false;
    return TSDF_OK;
}

// what both queue entry points share.  `fill` is null for page-locked plane buffers (copied from directly).
int queue_frame_common(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t width, int32_t height,
                       bool has_nrm, bool has_rgb, std::function<void(size_t, size_t)> fill) {
    int rc = bind_device(h);
    if (rc) return rc;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: a frame is queued already (the queue is two deep: current + next)");
    if (h->have_frame && (h->fw != width || h->fh != height))
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: the queued frame must have the size of the current one (%dx%d)", h->fw, h->fh);
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    tsdf_handle::Queued& q = h->queued;
    q.nb = h->fidx ^ 1; q.has_nrm = has_nrm; q.has_rgb = has_rgb; q.direct = !fill; q.device = false; q.err = hipSuccess; q.rc = TSDF_OK;
    h->staged_xyz = false;                   // in_xyz / in_nrm / in_rgb are about to hold the QUEUED frame, not the current one
    pick_pixel_layout(h, &q.su, &q.sv);      // from the pose of this moment: only the order of the records depends on it
    rc = wait_buffer_free(h, q.nb, h->fstream);
    if (rc) return rc;
    const PackArgs pa = pack_args(h, h->in_xyz, has_nrm ? h->in_nrm : nullptr, has_rgb ? h->in_rgb : nullptr, q.su, q.sv, q.nb);
    if (q.direct) {
        HIP_TRY(h, hipMemcpyAsync(h->in_xyz, xyz, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
        if (nrm) HIP_TRY(h, hipMemcpyAsync(h->in_nrm, nrm, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
        if (rgb) HIP_TRY(h, hipMemcpyAsync(h->in_rgb, rgb, npix * 3, hipMemcpyHostToDevice, h->fstream));
        HIP_TRY(h, hipEventRecord(h->ev_copied, h->fstream));
        HIP_TRY(h, launch_pack(h->fstream, pa));
        HIP_TRY(h, hipEventRecord(h->ev_queued, h->fstream));
        q.active = true;
        return TSDF_OK;
    }
    // pageable buffers: a library thread fills pinned staging planes (with the staging pool) and issues copies and pack
    // while the caller goes on.  Two sets of staging planes alternate: the one that is filled now last fed the copies of
    // the frame before the current one, and the library thread, not the caller, waits for those if it has to.
    rc = ensure_second_staging_set(h, npix);
    if (rc) return rc;
    {   // TSDF_STAGE_SETS=1 (diagnosis): wait here, on the caller's thread, as the single staging set of round 3 made it do
        static const bool one_set = [] { const char* e = std::getenv("TSDF_STAGE_SETS"); return e && std::atoi(e) == 1; }();
        if (one_set) HIP_TRY(h, hipStreamSynchronize(h->fstream));
    }
    const auto t_queued = std::chrono::steady_clock::now();
    if (!h->qthread.joinable()) {
        try { h->qthread = std::thread(queue_thread_main, h); }
        catch (...) { return fail(h, TSDF_E_NOMEM, "tsdf_queue_frame: cannot start the staging thread"); }
    }
    {
        std::lock_guard<std::mutex> g(h->qmu);
        h->qbusy = true;
        h->qjob = [h, npix, has_nrm, has_rgb, fill, pa, t_queued] {
            const auto ts0 = std::chrono::steady_clock::now();
            if (h->sp.on) h->sp.handoff += std::chrono::duration<double, std::nano>(ts0 - t_queued).count();
            // switch to the other staging set (fill and stage_and_upload read h->pin_* when they run)
            std::swap(h->pin_xyz, h->alt_xyz); std::swap(h->pin_nrm, h->alt_nrm); std::swap(h->pin_rgb, h->alt_rgb);
            std::swap(h->ev_stage_done[0], h->ev_stage_done[1]); std::swap(h->stage_recorded[0], h->stage_recorded[1]);
            hipError_t e = h->stage_recorded[0] ? hipEventSynchronize(h->ev_stage_done[0]) : hipSuccess;
            if (h->sp.on) h->sp.sync_before += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - ts0).count();
            if (e == hipSuccess) e = stage_and_upload(h, npix, true, has_nrm, has_rgb, fill);
            if (e == hipSuccess) { e = hipEventRecord(h->ev_stage_done[0], h->fstream); h->stage_recorded[0] = e == hipSuccess; }
            if (e == hipSuccess) e = launch_pack(h->fstream, pa);
            if (e == hipSuccess) e = hipEventRecord(h->ev_queued, h->fstream);
            h->queued.err = e;
        };
    }
    h->qcv.notify_all();
    q.active = true;
    return TSDF_OK;
}
}  // namespace

int tsdf_queue_frame(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t width, int32_t height) {
    if (!h || !xyz || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_queue_frame: bad argument") : TSDF_E_BADARG;
    const size_t npix = (size_t)width * height;
    const bool direct = is_pinned_host(xyz, npix * 12) && (!nrm || is_pinned_host(nrm, npix * 12)) && (!rgb || is_pinned_host(rgb, npix * 3));
    std::function<void(size_t, size_t)> fill;
    if (!direct) fill = [h, xyz, nrm, rgb](size_t i0, size_t i1) {
        std::memcpy(h->pin_xyz + 3 * i0, xyz + 3 * i0, (i1 - i0) * 3 * sizeof(float));
        if (nrm) std::memcpy(h->pin_nrm + 3 * i0, nrm + 3 * i0, (i1 - i0) * 3 * sizeof(float));
        if (rgb) std::memcpy(h->pin_rgb + 3 * i0, rgb + 3 * i0, (i1 - i0) * 3);
    };
    return queue_frame_common(h, xyz, nrm, rgb, width, height, nrm != nullptr, rgb != nullptr, fill);
}

int tsdf_queue_frame_aos(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L, int32_t width, int32_t height) {
    if (!h || !L || !points || width <= 0 || height <= 0)
        return h ? fail(h, TSDF_E_BADARG, "tsdf_queue_frame_aos: bad argument (the points are required)") : TSDF_E_BADARG;
    const bool color = L->r_offset >= 0 && L->g_offset >= 0 && L->b_offset >= 0;
    if (L->point_stride < 12 || L->xyz_offset < 0 || L->xyz_offset + 12 > L->point_stride ||
        (color && (L->r_offset >= L->point_stride || L->g_offset >= L->point_stride || L->b_offset >= L->point_stride)))
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame_aos: point layout (stride %d, xyz at %d) does not hold three floats and the colour bytes",
                    L->point_stride, L->xyz_offset);
    if (normals && (L->normal_stride < 12 || L->normal_offset < 0 || L->normal_offset + 12 > L->normal_stride))
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame_aos: normal layout (stride %d, normal at %d) does not hold three floats",
                    L->normal_stride, L->normal_offset);
    const tsdf_aos_layout lay = *L;
    std::function<void(size_t, size_t)> fill = [h, points, normals, lay, color](size_t i0, size_t i1) {
        repack_aos(lay, points, normals, color, h->pin_xyz, h->pin_nrm, h->pin_rgb, i0, i1);
    };
    return queue_frame_common(h, nullptr, nullptr, nullptr, width, height, normals != nullptr, color, fill);
}

int tsdf_queue_frame_device(tsdf_handle* h, const float* d_xyz, const float* d_nrm, const uint8_t* d_rgb, int32_t width, int32_t height) {
    if (!h || !d_xyz || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_queue_frame_device: bad argument") : TSDF_E_BADARG;
    int rc = bind_device(h);
    if (rc) return rc;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: a frame is queued already (the queue is two deep: current + next)");
    if (h->have_frame && (h->fw != width || h->fh != height))
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: the queued frame must have the size of the current one (%dx%d)", h->fw, h->fh);
    rc = ensure_frame_buffers(h, width, height, false);
    if (rc) return rc;
    tsdf_handle::Queued& q = h->queued;
    q.nb = h->fidx ^ 1; q.has_nrm = d_nrm != nullptr; q.has_rgb = d_rgb != nullptr; q.direct = true; q.device = true; q.err = hipSuccess; q.rc = TSDF_OK;
    q.deferred = q.packed = false;
    if (h->defer_device_pack) {
        // no launch now: the integrate launch of the CURRENT frame packs this one in workgroups appended to its
        // list_rows_kernel (tsdf_integrate), on the main stream, i.e. behind the last reader of the record buffer
        q.deferred = true; q.d_xyz = d_xyz; q.d_nrm = d_nrm; q.d_rgb = d_rgb;
        q.active = true;
        borrow_device_frame(h, h->frame_serial + 1);
        return TSDF_OK;
    }
    pick_pixel_layout(h, &q.su, &q.sv);
    // the record buffer of the frame before the current one: free once that frame's integration is done -- from then
    // on the pack runs on the frame stream, next to the current frame's tracker passes
    rc = wait_buffer_free(h, q.nb, h->fstream);
    if (rc) return rc;
    HIP_TRY(h, launch_pack(h->fstream, pack_args(h, d_xyz, d_nrm, d_rgb, q.su, q.sv, q.nb)));
    borrow_device_frame(h, h->frame_serial + 1);
    HIP_TRY(h, launch_release(h->fstream, release_for(h, h->frame_serial + 1, 1)));
    HIP_TRY(h, hipEventRecord(h->ev_queued, h->fstream));
    q.active = true;
    return TSDF_OK;
}

int tsdf_next_frame(tsdf_handle* h) {
    if (!h) return TSDF_E_BADARG;
    tsdf_handle::Queued& q = h->queued;
    if (!q.active) return fail(h, TSDF_E_NO_FRAME, "tsdf_next_frame: no frame is queued");
    int rc = bind_device(h);
    if (rc) return rc;
    q.active = false;
    const bool from_device = q.device;
    q.device = false;
    if (from_device && q.deferred) {
        q.deferred = false;
        h->staged_xyz = false;
        if (!q.packed) return defer_pack(h, q.d_xyz, q.d_nrm, q.d_rgb, true);      // no integrate launch came by: as tsdf_set_frame_device
        // packed inside the previous frame's integrate launch, on the main stream: nothing to wait for
        if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);
        h->fidx = q.nb; h->pn = h->pn_buf[q.nb]; h->samples = h->samples_buf[q.nb];
        h->deferred = tsdf_handle::DeferredPack();
        h->pix_su = q.su; h->pix_sv = q.sv;
        h->frame_side = false;
        h->have_frame = true;
        h->frame_serial++;
        h->frame_has_nrm = q.has_nrm;
        h->frame_has_rgb = q.has_rgb;
        return TSDF_OK;
    }
    if (from_device) {
        // nothing to wait for on the host: device buffers stay borrowed as tsdf_set_frame_device's do
    } else if (q.direct) {
        HIP_TRY(h, hipEventSynchronize(h->ev_copied));       // the caller's buffers have been read
    } else {
        const auto tw0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> g(h->qmu);
        h->qcv.wait(g, [&] { return !h->qbusy; });           // the staging thread is done with the caller's buffers
        if (h->sp.on) h->sp.next_wait += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tw0).count();
        if (q.rc != TSDF_OK) { const int r = q.rc; q.rc = TSDF_OK; h->err = q.msg; return r; }
        if (q.err != hipSuccess) return fail(h, TSDF_E_HIP, "tsdf_queue_frame: staging failed: %s", hipGetErrorString(q.err));
    }
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_queued, 0));   // everything queued on `stream` from here on sees the frame
    if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);     // the frame this one replaces was never packed
    h->fidx = q.nb; h->pn = h->pn_buf[q.nb]; h->samples = h->samples_buf[q.nb];
    h->deferred = tsdf_handle::DeferredPack();
    h->pix_su = q.su; h->pix_sv = q.sv;
    h->frame_side = true;
    h->have_frame = true;
    h->staged_xyz = !from_device;
    h->frame_serial++;
    h->frame_has_nrm = q.has_nrm;
    h->frame_has_rgb = q.has_rgb;
    return TSDF_OK;
}

int tsdf_set_frame_device(tsdf_handle* h, const float* d_xyz, const float* d_nrm, const uint8_t* d_rgb, int32_t width, int32_t height) {
    if (!h || !d_xyz || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_set_frame_device: bad argument") : TSDF_E_BADARG;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_set_frame_device");
    int rc = bind_device(h);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, false);
    if (rc) return rc;
    h->staged_xyz = false;
    if (h->defer_device_pack) return defer_pack(h, d_xyz, d_nrm, d_rgb);
    return run_pack(h, d_xyz, d_nrm, d_rgb, h->stream, true);
}

int tsdf_set_frame_aos(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L,
                       int32_t width, int32_t height) {
    if (!h || !L || (!points && !normals) || width <= 0 || height <= 0)
        return h ? fail(h, TSDF_E_BADARG, "tsdf_set_frame_aos: bad argument") : TSDF_E_BADARG;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_set_frame_aos");
    const bool color = points && L->r_offset >= 0 && L->g_offset >= 0 && L->b_offset >= 0;
    if (points && (L->point_stride < 12 || L->xyz_offset < 0 || L->xyz_offset + 12 > L->point_stride ||
                   (color && (L->r_offset >= L->point_stride || L->g_offset >= L->point_stride || L->b_offset >= L->point_stride))))
        return fail(h, TSDF_E_BADARG, "tsdf_set_frame_aos: point layout (stride %d, xyz at %d) does not hold three floats and the colour bytes",
                    L->point_stride, L->xyz_offset);
    if (normals && (L->normal_stride < 12 || L->normal_offset < 0 || L->normal_offset + 12 > L->normal_stride))
        return fail(h, TSDF_E_BADARG, "tsdf_set_frame_aos: normal layout (stride %d, normal at %d) does not hold three floats",
                    L->normal_stride, L->normal_offset);
    if (!points && !(h->have_frame && h->staged_xyz && h->fw == width && h->fh == height))
        return fail(h, TSDF_E_NO_FRAME, "tsdf_set_frame_aos: normals alone complete the CURRENT host frame of the same size; there is none");
    int rc = bind_device(h);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    HIP_TRY(h, hipStreamSynchronize(h->fstream));          // the pinned staging buffers may still feed the previous frame
    const bool had_rgb = h->frame_has_rgb;
    h->staged_xyz = false;                                 // until this frame's planes are complete on the device
    float* const px = h->pin_xyz; float* const pnm = h->pin_nrm; uint8_t* const pc = h->pin_rgb;
    const tsdf_aos_layout lay = *L;
    // a new cloud: its tracker samples go up first (the passes of a following tsdf_track run under the planes' copy)
    static const bool samples_first_on = [] { const char* e = std::getenv("TSDF_SAMPLES_FIRST"); return !(e && std::atoi(e) == 0); }();
    const bool samples_first = samples_first_on && points != nullptr;
    if (samples_first) { rc = upload_samples_first(h, points, (size_t)lay.point_stride, (size_t)lay.xyz_offset, width); if (rc) return rc; }
    HIP_TRY(h, stage_and_upload(h, npix, points != nullptr, normals != nullptr, color, [&](size_t i0, size_t i1) {
        repack_aos(lay, points, normals, color, px, pnm, pc, i0, i1);
    }));
    const bool has_rgb = points ? color : had_rgb;
    h->staged_xyz = true;
    return run_pack(h, h->in_xyz, normals ? h->in_nrm : nullptr, has_rgb ? h->in_rgb : nullptr, h->fstream, false, samples_first);
}

// ---- depth pre-processing (optional stage in front of the hot path) -------------------------------------------

void tsdf_default_preproc(tsdf_preproc_params* p) {
    if (!p) return;
    p->depth_scale = 1.0f / 5000.0f;
    p->sigma_s = 15.0f;
    p->sigma_r = 0.05f;
    p->radius = 30;
    p->normal_radius = 5;
    p->max_depth_change = 0.02f;
    p->grid_filter = 1;
}

namespace {
// argument checks and buffers shared by tsdf_set_depth_frame / tsdf_queue_depth_frame (caller's thread)
int depth_frame_prepare(tsdf_handle* h, const char* who, bool queued, const uint16_t* depth16, const float* depthf, int32_t width,
                        int32_t height, const tsdf_preproc_params* params, tsdf_preproc_params* pp_out) {
    if (!h || (!depth16 == !depthf) || width <= 0 || height <= 0)
        return h ? fail(h, TSDF_E_BADARG, "%s: exactly one of depth16 / depthf, positive size", who) : TSDF_E_BADARG;
    if (!h->have_K) return fail(h, TSDF_E_NO_INTRINSICS, "%s needs the intrinsics for the back-projection", who);
    tsdf_preproc_params pp;
    if (params) pp = *params; else tsdf_default_preproc(&pp);
    if (pp.radius < 0 || pp.radius > 32 || pp.normal_radius < 1 || pp.normal_radius > 8 || !(pp.sigma_s > 0) || !(pp.sigma_r > 0))
        return fail(h, TSDF_E_BADARG, "%s: bad parameters (radius %d, normal_radius %d)", who, pp.radius, pp.normal_radius);
    const bool use_grid = pp.grid_filter != 0 && pp.radius > 0;
    if (use_grid && !(pp.sigma_s >= 1.0f && pp.sigma_s <= 30.0f))
        return fail(h, TSDF_E_BADARG, "%s: the bilateral grid takes sigma_s in [1, 30] pixels, not %g", who, (double)pp.sigma_s);
    if (depth16 && !(pp.depth_scale > 0))
        return fail(h, TSDF_E_BADARG, "%s: depth_scale must be positive", who);
    if (h->queued.active)
        return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", who);
    int rc = bind_device(h);
    if (rc) return rc;
    if (queued && h->have_frame && (h->fw != width || h->fh != height))
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: the queued frame must have the size of the current one (%dx%d)", h->fw, h->fh);
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    if (npix > h->pre_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->fstream));
        free_preproc(h);
        HIP_TRY(h, hipMalloc((void**)&h->pre_z, npix * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->pre_zf, npix * sizeof(float)));
        HIP_TRY(h, hipMalloc((void**)&h->pre_depth, npix * sizeof(float)));
        HIP_TRY(h, hipHostMalloc((void**)&h->pin_depth, npix * sizeof(float), hipHostMallocDefault));
        HIP_TRY(h, hipMalloc((void**)&h->pre_minmax, 2 * sizeof(unsigned)));
        HIP_TRY(h, hipHostMalloc((void**)&h->pin_minmax, 2 * sizeof(unsigned), hipHostMallocDefault));
        h->pre_cap = npix;
    }
    *pp_out = pp;
    return TSDF_OK;
}

// Upload, back-projection, filter and normals of a depth frame on the frame stream; in_xyz / in_nrm / in_rgb hold the
// frame afterwards.  Runs on the caller's thread (tsdf_set_depth_frame) or on the queue's library thread
// (tsdf_queue_depth_frame): the bilateral grid's depth extent is the one host round trip of this path.
int depth_frame_work(tsdf_handle* h, const char* who, const uint16_t* depth16, const float* depthf, const uint8_t* rgb,
                     int32_t width, int32_t height, const tsdf_preproc_params& pp, bool* direct_out) {
    const size_t npix = (size_t)width * height;
    const bool use_grid = pp.grid_filter != 0 && pp.radius > 0;
    HIP_TRY(h, hipStreamSynchronize(h->fstream));          // pinned staging may still feed the previous frame
    const size_t dbytes = npix * (depth16 ? sizeof(uint16_t) : sizeof(float));
    const void* dsrc = depth16 ? (const void*)depth16 : (const void*)depthf;
    // page-locked caller buffers are copied from directly, as in tsdf_set_frame
    const bool direct = is_pinned_host(dsrc, dbytes) && (!rgb || is_pinned_host(rgb, npix * 3));
    *direct_out = direct;
    if (!direct) std::memcpy(h->pin_depth, dsrc, dbytes);
    HIP_TRY(h, hipMemcpyAsync(h->pre_depth, direct ? dsrc : h->pin_depth, dbytes, hipMemcpyHostToDevice, h->fstream));
    HIP_TRY(h, launch_depth_to_z(h->fstream, depth16 ? (const uint16_t*)h->pre_depth : nullptr,
                                 depth16 ? nullptr : (const float*)h->pre_depth, pp.depth_scale, (int)npix, h->pre_z,
                                 use_grid ? h->pre_minmax : nullptr));
    if (use_grid)
        HIP_TRY(h, hipMemcpyAsync(h->pin_minmax, h->pre_minmax, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, h->fstream));
    if (rgb) {
        if (!direct) std::memcpy(h->pin_rgb, rgb, npix * 3);
        HIP_TRY(h, hipMemcpyAsync(h->in_rgb, direct ? rgb : h->pin_rgb, npix * 3, hipMemcpyHostToDevice, h->fstream));
    }
    if (direct && !use_grid) HIP_TRY(h, hipEventRecord(h->ev_copied, h->fstream));
    // The grid's depth extent follows the frame's depth range: the one host round trip of this path (8 bytes).
    BilateralGrid bg;
    bool grid_on = false;
    if (use_grid) {
        HIP_TRY(h, hipStreamSynchronize(h->fstream));
        if (h->pin_minmax[0] != 0xffffffffu) {             // else no valid pixel at all: nothing to filter
            float zmin, zmax;
            const unsigned lo = h->pin_minmax[0], hi = ~h->pin_minmax[1];
            std::memcpy(&zmin, &lo, 4); std::memcpy(&zmax, &hi, 4);
            if (!bilateral_grid_plan(width, height, pp.sigma_s, pp.sigma_r, zmin, zmax, &bg))
                return fail(h, TSDF_E_BADARG, "%s: depth range [%g, %g] m is not a usable bilateral grid at sigma_r %g", who,
                            (double)zmin, (double)zmax, (double)pp.sigma_r);
            const size_t cells = (size_t)bg.gx * bg.gy * bg.gz;
            if (cells > ((size_t)1 << 26))
                return fail(h, TSDF_E_BADARG, "%s: bilateral grid of %d x %d x %d cells is too large (sigma_s %g, sigma_r %g)", who,
                            bg.gx, bg.gy, bg.gz, (double)pp.sigma_s, (double)pp.sigma_r);
            if (cells > h->pre_grid_cap) {
                if (h->pre_grid_a) (void)hipFree(h->pre_grid_a);
                if (h->pre_grid_b) (void)hipFree(h->pre_grid_b);
                h->pre_grid_a = h->pre_grid_b = nullptr; h->pre_grid_cap = 0;
                const size_t cap = cells + cells / 2;      // the range moves from frame to frame: head-room
                HIP_TRY(h, hipMalloc((void**)&h->pre_grid_a, cap * sizeof(float2)));
                HIP_TRY(h, hipMalloc((void**)&h->pre_grid_b, cap * sizeof(float2)));
                h->pre_grid_cap = cap;
            }
            grid_on = true;
        }
    }
    const float Kf[4] = {(float)h->K[0], (float)h->K[4], (float)h->K[2], (float)h->K[5]};
    HIP_TRY(h, launch_preproc(h->fstream, width, height, Kf, use_grid && !grid_on ? 0 : pp.radius, pp.sigma_s, pp.sigma_r,
                              pp.normal_radius, pp.max_depth_change, grid_on ? &bg : nullptr, h->pre_grid_a, h->pre_grid_b,
                              h->pre_z, h->pre_zf, h->in_xyz, h->in_nrm));
    return TSDF_OK;
}
}  // namespace

int tsdf_set_depth_frame(tsdf_handle* h, const uint16_t* depth16, const float* depthf, const uint8_t* rgb,
                         int32_t width, int32_t height, const tsdf_preproc_params* params) {
    tsdf_preproc_params pp;
    int rc = depth_frame_prepare(h, "tsdf_set_depth_frame", false, depth16, depthf, width, height, params, &pp);
    if (rc) return rc;
    h->staged_xyz = false;                                 // until this frame's planes are complete on the device
    bool direct = false;
    rc = depth_frame_work(h, "tsdf_set_depth_frame", depth16, depthf, rgb, width, height, pp, &direct);
    if (rc) return rc;
    h->staged_xyz = true;
    rc = run_pack(h, h->in_xyz, h->in_nrm, rgb ? h->in_rgb : nullptr, h->fstream);
    if (rc) return rc;
    const bool use_grid = pp.grid_filter != 0 && pp.radius > 0;
    if (direct && !use_grid) HIP_TRY(h, hipEventSynchronize(h->ev_copied));   // (the grid path has synchronised already)
    return TSDF_OK;
}

// The two-deep queue for raw depth frames: upload, pre-processing (with its one host round trip for the bilateral grid's
// depth range) and packing of frame k+1 run on the library thread + frame stream while the caller drives frame k's
// Gauss-Newton passes; the buffers are borrowed until tsdf_next_frame returns.
int tsdf_queue_depth_frame(tsdf_handle* h, const uint16_t* depth16, const float* depthf, const uint8_t* rgb,
                           int32_t width, int32_t height, const tsdf_preproc_params* params) {
    tsdf_preproc_params pp;
    int rc = depth_frame_prepare(h, "tsdf_queue_depth_frame", true, depth16, depthf, width, height, params, &pp);
    if (rc) return rc;
    tsdf_handle::Queued& q = h->queued;
    q.nb = h->fidx ^ 1; q.has_nrm = true; q.has_rgb = rgb != nullptr; q.direct = false; q.device = false; q.err = hipSuccess;
    q.deferred = q.packed = false; q.rc = TSDF_OK;
    h->staged_xyz = false;                   // in_xyz / in_nrm / in_rgb are about to hold the QUEUED frame, not the current one
    pick_pixel_layout(h, &q.su, &q.sv);
    rc = wait_buffer_free(h, q.nb, h->fstream);
    if (rc) return rc;
    const PackArgs pa = pack_args(h, h->in_xyz, h->in_nrm, rgb ? h->in_rgb : nullptr, q.su, q.sv, q.nb);
    if (!h->qthread.joinable()) {
        try { h->qthread = std::thread(queue_thread_main, h); }
        catch (...) { return fail(h, TSDF_E_NOMEM, "tsdf_queue_depth_frame: cannot start the staging thread"); }
    }
    {
        std::lock_guard<std::mutex> g(h->qmu);
        h->qbusy = true;
        h->qjob = [h, depth16, depthf, rgb, width, height, pp, pa] {
            bool direct = false;
            t_err_sink = &h->queued.msg;
            int r = depth_frame_work(h, "tsdf_queue_depth_frame", depth16, depthf, rgb, width, height, pp, &direct);
            t_err_sink = nullptr;
            hipError_t e = hipSuccess;
            if (r == TSDF_OK) e = launch_pack(h->fstream, pa);
            if (r == TSDF_OK && e == hipSuccess) e = hipEventRecord(h->ev_queued, h->fstream);
            const bool use_grid = pp.grid_filter != 0 && pp.radius > 0;
            if (r == TSDF_OK && e == hipSuccess && direct && !use_grid) e = hipEventSynchronize(h->ev_copied);   // the caller's buffers have been read
            h->queued.rc = r;
            h->queued.err = e;
        };
    }
    h->qcv.notify_all();
    q.active = true;
    return TSDF_OK;
}

int tsdf_get_preprocessed(tsdf_handle* h, float* xyz, float* nrm) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!h->in_xyz || h->in_cap < (size_t)h->fw * h->fh) return fail(h, TSDF_E_NO_FRAME, "no pre-processed frame held");
    // the staging planes hold the CURRENT frame only until the next frame is queued (tsdf_queue_frame / _aos /
    // tsdf_queue_depth_frame fill them with frame k+1, on the library thread and the frame stream, while frame k is current)
    if (!h->staged_xyz)
        return fail(h, TSDF_E_NO_FRAME, "tsdf_get_preprocessed: the planes of the current frame are no longer held (a frame is queued behind "
                                        "it, or the frame came from device memory): read them before queueing the next frame");
    const size_t bytes = (size_t)h->fw * h->fh * 3 * sizeof(float);
    if (xyz) HIP_TRY(h, hipMemcpyAsync(xyz, h->in_xyz, bytes, hipMemcpyDeviceToHost, h->fstream));
    if (nrm) HIP_TRY(h, hipMemcpyAsync(nrm, h->in_nrm, bytes, hipMemcpyDeviceToHost, h->fstream));
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    return TSDF_OK;
}

// ---- hot path ----------------------------------------------------------------------------------------

int tsdf_integrate(tsdf_handle* h, tsdf_integrate_stats* stats) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!h->have_K) return fail(h, TSDF_E_NO_INTRINSICS, "camera matrix not received (reference: sdf.cpp:227-230 exits)");
    if (!h->frame_has_nrm) return fail(h, TSDF_E_NO_FRAME, "tsdf_integrate needs normals in the current frame");
    if (h->cfg.with_color && !h->frame_has_rgb)
        return fail(h, TSDF_E_NO_FRAME, "with_color=1 needs rgb in the current frame");
    if (h->records_pending) {                // samples-first frame: the records come off the frame stream behind the planes' copy
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_frame, 0));
        h->records_pending = false;
    }
    if (h->deferred.pending) choose_pixel_layout(h);      // the records are written in this launch: lay them out for the pose they are read at
    IntegrateParams p;
    fill_integrate_params(h, p);
    unsigned long long before[kNumCounters];
    if (stats) {
        rc = fetch_counters(h);
        if (rc) return rc;
        std::memcpy(before, h->counters_host, sizeof before);
    }
    EventPair* ep;
    rc = timed_begin(h, 0, &ep, h->stream);
    if (rc) return rc;
    // Deferred packing: one frame's records can be written inside this launch (workgroups appended to list_rows_kernel).
    // A queued device frame goes first -- that is the packing of the NEXT frame, sample list included, hidden under this
    // frame's list kernel; the current frame's own records then need a launch in front (the first frame of a stream only).
    PackArgs pa;
    bool fused = false, fused_queued = false;
    ReleaseWord rel;                         // tells the host when the borrowed planes packed by this launch have been read
    tsdf_handle::Queued& q = h->queued;
    if (q.active && q.device && q.deferred && !q.packed) {
        const bool own_too = h->deferred.pending;
        if (own_too) {
            PackArgs own = pack_args(h, h->deferred.xyz, h->deferred.nrm, h->deferred.rgb, h->pix_su, h->pix_sv, h->fidx);
            if (h->deferred.samples_listed) own.samples = nullptr;
            HIP_TRY(h, launch_pack(h->stream, own));
            h->deferred.pending = false;
        }
        q.su = h->pix_su; q.sv = h->pix_sv;      // laid out for this frame's pose: the next one's is close to it
        pa = pack_args(h, q.d_xyz, q.d_nrm, q.d_rgb, q.su, q.sv, q.nb);
        rel = release_for(h, h->frame_serial + 1, 0);
        if (own_too)                             // the launch in front, same stream: read by the time the ticket appears
            for (auto& b : h->borrowed) if (b.serial == h->frame_serial) { b.stream = 0; b.ticket = rel.ticket; }
        fused = fused_queued = true;
    } else if (h->deferred.pending) {
        pa = pack_args(h, h->deferred.xyz, h->deferred.nrm, h->deferred.rgb, h->pix_su, h->pix_sv, h->fidx);
        if (h->deferred.samples_listed) pa.samples = nullptr;     // a tracker pass has written them already
        rel = release_for(h, h->frame_serial, 0);
        fused = true;
    }
    {
        // Grid by work: a persistent workgroup costs ~2.5 us of launch per workgroup and CU whatever it finds to do (segment
        // table, k table, pipeline fill and drain), so the grid follows the work -- enough workgroups per CU that a
        // wavefront gets >= 16 items, going by the LAST launch's item count (consecutive frames list nearly the same rows;
        // the count arrives in pinned memory, nothing waits for it).  A whole 512^3 volume lists ~195 k items and keeps
        // five workgroups per CU; the 1/8 slab of an 8-GPU job lists ~24 k and gets two.  Only the schedule changes.
        int blocks = h->integrate_blocks;
        const unsigned long long last_items = __atomic_load_n(h->release_host + 2, __ATOMIC_RELAXED);
        if (h->integrate_grid_by_work && last_items != ~0ull && h->integrate_cus > 0) {
            const unsigned long long per_wg_cu = (unsigned long long)h->integrate_cus * (kIntegrateBlock / 64) * 16ull;   // items that give every wavefront 16
            const int max_per_cu = h->integrate_blocks / ((h->integrate_cus + 7) / 8 * 8) > 0 ? h->integrate_blocks / ((h->integrate_cus + 7) / 8 * 8) : 1;
            int want = (int)((last_items + per_wg_cu - 1) / per_wg_cu);
            want = want < 1 ? 1 : want > max_per_cu ? max_per_cu : want;
            blocks = (h->integrate_cus * want + 7) / 8 * 8;
            if (blocks > h->integrate_blocks) blocks = h->integrate_blocks;
        }
        rel.items_word = h->release_host + 2;
        const hipError_t le = launch_integrate(h->stream, p, h->dw, h->crgb, h->pn, h->counters, h->worklist, h->work_count,
                                               blocks, h->integrate_launches, h->wg_counts, h->integrate_queue,
                                               fused ? &pa : nullptr, &rel);
        if (le != hipSuccess) {
            // nothing was packed: the frames stay borrowed and unpacked (a later launch, or tsdf_synchronize, packs them)
            if (fused) for (auto& b : h->borrowed) if (b.stream == 0 && b.ticket == rel.ticket && b.serial >= h->frame_serial + (fused_queued ? 1 : 0)) b.stream = -1;
            return fail(h, TSDF_E_HIP, "launch_integrate failed: %s (%s:%d)", hipGetErrorString(le), __FILE__, __LINE__);
        }
    }
    h->integrate_launches++;
    if (fused_queued) q.packed = true;       // only now: a failed launch must not leave an unpacked record buffer marked as packed
    h->deferred.pending = false;             // records and sample list of the current frame are complete from here on
    rc = timed_end(h, ep, h->stream);
    if (rc) return rc;
    if (h->frame_side) {     // the next-but-one pack (on the frame stream) may overwrite this buffer after this launch
        HIP_TRY(h, hipEventRecord(h->ev_buf_used[h->fidx], h->stream));
        h->used_valid[h->fidx] = true;
    } else {                 // frames packed on the main stream are ordered by the stream itself: no event per frame
        h->used_valid[h->fidx] = false;
        h->used_untracked[h->fidx] = true;
    }
    h->cnt.integrate_calls++;
    h->cnt.n_voxels_swept += h->n_stored;
    if (stats) {
        rc = fetch_counters(h);
        if (rc) return rc;
        stats->n_updated = (int64_t)(h->counters_host[kCntUpdatedOwned] - before[kCntUpdatedOwned]);
        stats->n_updated_halo = (int64_t)(h->counters_host[kCntUpdatedHalo] - before[kCntUpdatedHalo]);
        stats->n_voxels = h->n_stored;
    }
    return TSDF_OK;
}

int tsdf_accumulate(tsdf_handle* h, double A[36], double b[6], tsdf_accum_stats* stats) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    if (!A || !b) return fail(h, TSDF_E_BADARG, "tsdf_accumulate: null output");
    rc = accumulate_pass(h, false);
    if (rc) return rc;
    unpack_normal_equations(h->red_host, A, b);
    if (stats) {
        stats->n_samples = (int64_t)h->red_host[33];
        stats->n_nan = (int64_t)h->red_host[32];
        stats->n_oog = (int64_t)h->red_host[31];
        stats->n_in_grid_owned = (int64_t)h->red_host[30];
        stats->n_ok = (int64_t)h->red_host[29];
        stats->n_terms = (int64_t)h->red_host[27];
    }
    return TSDF_OK;
}

int tsdf_gn_update(tsdf_handle* h, const double A[36], const double b[6], double twist[6], int32_t* stop) {
    if (!h || !A || !b) return TSDF_E_BADARG;
    double tw[6];
    bool st = false;
    if (!hm::gn_step(h->pose, A, b, h->cfg.max_twist_diff, tw, &st))
        return fail(h, TSDF_E_SINGULAR, "normal equations singular or pose not finite; pose left unchanged");
    if (twist) std::memcpy(twist, tw, sizeof tw);
    if (stop) *stop = st ? 1 : 0;
    return TSDF_OK;
}

namespace {
int track_loop(tsdf_handle* h, tsdf_track_stats* stats);
}
int tsdf_track(tsdf_handle* h, tsdf_track_stats* stats) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    return track_loop(h, stats);
}
namespace {
// camera_tracking.cpp:79-239: the Gauss-Newton loop on the current frame's sample list
int track_loop(tsdf_handle* h, tsdf_track_stats* stats) {
    int rc = TSDF_OK;
    bool stop = false;
    int g = 0;
    double A[36], b[6], twist[6] = {0, 0, 0, 0, 0, 0};
    int64_t n_terms = 0;
    h->cnt.track_calls++;
    // The pose advances in place pass by pass (camera_tracking.cpp:237-239).  Whatever goes wrong in a later pass,
    // the caller gets the pose it came in with: a half-converged pose is never left behind (tsdf.h: TSDF_E_SINGULAR /
    // TSDF_E_NO_SAMPLES / TSDF_E_HALO / TSDF_E_COMM / TSDF_E_HIP "pose left unchanged").
    const hm::Pose entry = h->pose;
    for (g = 0; g < h->cfg.gn_max_iter && !stop; ++g) {            // camera_tracking.cpp:79
        const auto tq0 = h->track_profile ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        const double waited = h->tp_wait, before = h->tp_fill + h->tp_launch;
        rc = accumulate_pass(h, true, g > 0);
        if (rc) { h->pose = entry; return rc; }
        n_terms = (int64_t)h->red_host[27];
        if (n_terms == 0) {
            h->pose = entry;
            return fail(h, TSDF_E_NO_SAMPLES, "no valid tracking sample (iteration %d); pose left unchanged", g);
        }
        unpack_normal_equations(h->red_host, A, b);
        if (!hm::gn_step(h->pose, A, b, h->cfg.max_twist_diff, twist, &stop)) {
            h->pose = entry;
            return fail(h, TSDF_E_SINGULAR, "normal equations singular at iteration %d; pose left unchanged", g);
        }
        if (h->track_profile)      // everything of this pass that was neither parameters, launch nor waiting
            h->tp_post += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tq0).count() -
                          (h->tp_wait - waited) - (h->tp_fill + h->tp_launch - before);
    }
    if (stats) {
        stats->iterations = g;
        stats->stopped = stop ? 1 : 0;
        stats->n_terms_last = n_terms;
        std::memcpy(stats->last_twist, twist, sizeof twist);
    }
    return TSDF_OK;
}
}  // namespace

int tsdf_track_and_integrate(tsdf_handle* h, int32_t do_track, tsdf_track_stats* track_stats, tsdf_integrate_stats* integrate_stats) {
    if (do_track) {
        const int rc = tsdf_track(h, track_stats);
        if (rc) return rc;
    }
    return tsdf_integrate(h, integrate_stats);
}

// ---- the reference's two hot calls on its own clouds (sdf_reconstruction.cpp:70,74) -------------------------------------
// kinect_callback calls estimate_new_position(sdf, cloud) and then update(tracker, cloud, normals), synchronously, with the
// clouds in pageable memory.  Through tsdf_set_frame_aos that was: upload ALL points (repack 9.8 MB, copy 4.6 MB, pack) ->
// track -> wait for the frame stream on the host, upload the normals, pack again -> integrate.  The tracker needs 34 240 of
// the 307 200 points: tsdf_track_aos gathers those into a pinned list (0.5 MB), copies it, and starts the Gauss-Newton
// passes; the whole cloud is repacked by the library threads and copied on the frame stream UNDER the passes.  The cloud
// is the caller's again when the call returns (the repack is over; the copy reads the library's pinned planes).
// tsdf_integrate_aos adds the normals (repack, one copy), checks -- under that copy -- that `points` still is the cloud
// that was tracked, byte for byte, and only uploads it again when it is not.
namespace {
int check_point_layout(tsdf_handle* h, const char* who, const tsdf_aos_layout* L, bool* color) {
    *color = L->r_offset >= 0 && L->g_offset >= 0 && L->b_offset >= 0;
    if (L->point_stride < 12 || L->xyz_offset < 0 || L->xyz_offset + 12 > L->point_stride ||
        (*color && (L->r_offset >= L->point_stride || L->g_offset >= L->point_stride || L->b_offset >= L->point_stride)))
        return fail(h, TSDF_E_BADARG, "%s: point layout (stride %d, xyz at %d) does not hold three floats and the colour bytes", who,
                    L->point_stride, L->xyz_offset);
    return TSDF_OK;
}
int check_normal_layout(tsdf_handle* h, const char* who, const tsdf_aos_layout* L) {
    if (L->normal_stride < 12 || L->normal_offset < 0 || L->normal_offset + 12 > L->normal_stride)
        return fail(h, TSDF_E_BADARG, "%s: normal layout (stride %d, normal at %d) does not hold three floats", who,
                    L->normal_stride, L->normal_offset);
    return TSDF_OK;
}
// do the points [i0, i1) of an array-of-structs cloud still hold the bytes that were repacked into the planes?
bool points_equal_planes(const tsdf_aos_layout& lay, const void* points, bool color, const float* px, const uint8_t* pc, size_t i0, size_t i1) {
    const char* p = (const char*)points + i0 * (size_t)lay.point_stride;
    unsigned diff = 0u;
    for (size_t i = i0; i < i1; ++i, p += lay.point_stride) {
        diff |= (unsigned)(std::memcmp(px + 3 * i, p + lay.xyz_offset, 12) != 0);
        if (color) diff |= (unsigned)((uint8_t)p[lay.r_offset] ^ pc[3 * i]) | (unsigned)((uint8_t)p[lay.g_offset] ^ pc[3 * i + 1]) | (unsigned)((uint8_t)p[lay.b_offset] ^ pc[3 * i + 2]);
    }
    return diff == 0u;
}
bool normals_equal_plane(const tsdf_aos_layout& lay, const void* normals, const float* pnm, size_t i0, size_t i1) {
    const char* p = (const char*)normals + i0 * (size_t)lay.normal_stride + lay.normal_offset;
    unsigned diff = 0u;
    for (size_t i = i0; i < i1; ++i, p += lay.normal_stride) diff |= (unsigned)(std::memcmp(pnm + 3 * i, p, 12) != 0);
    return diff == 0u;
}
int track_aos_impl(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L, int32_t width, int32_t height, tsdf_track_stats* stats);
}  // namespace

int tsdf_track_aos(tsdf_handle* h, const void* points, const tsdf_aos_layout* L, int32_t width, int32_t height, tsdf_track_stats* stats) {
    return track_aos_impl(h, points, nullptr, L, width, height, stats);
}
int tsdf_track_frame_aos(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L, int32_t width, int32_t height,
                         tsdf_track_stats* stats) {
    if (h && !normals) return fail(h, TSDF_E_BADARG, "tsdf_track_frame_aos: the normals are required (tsdf_track_aos takes the points alone)");
    return track_aos_impl(h, points, normals, L, width, height, stats);
}

namespace {
int track_aos_impl(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L, int32_t width, int32_t height, tsdf_track_stats* stats) {
    if (!h || !L || !points || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_track_aos: bad argument") : TSDF_E_BADARG;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_track_aos");
    bool color = false;
    int rc = check_point_layout(h, "tsdf_track_aos", L, &color);
    if (rc) return rc;
    if (normals) { rc = check_normal_layout(h, "tsdf_track_frame_aos", L); if (rc) return rc; }
    using pclk = std::chrono::steady_clock;
    const bool prof = h->sp.on;
    auto lap = [prof](pclk::time_point& t, double& acc) { if (prof) { const pclk::time_point n = pclk::now(); acc += std::chrono::duration<double, std::nano>(n - t).count(); t = n; } };
    pclk::time_point tp = prof ? pclk::now() : pclk::time_point();
    static const bool plain = [] { const char* e = std::getenv("TSDF_TRACK_AOS"); return e && std::atoi(e) == 0; }();
    if (plain) {                                 // TSDF_TRACK_AOS=0 (comparison): the whole cloud in front of the passes, as round 4's shim did
        rc = tsdf_set_frame_aos(h, points, normals, L, width, height);
        return rc ? rc : tsdf_track(h, stats);
    }
    rc = bind_device(h);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    rc = ensure_second_staging_set(h, npix);
    if (rc) return rc;
    lap(tp, h->sp.a_prep1);
    rc = ensure_pin_samples(h);
    if (rc) return rc;
    if (!h->qthread.joinable()) {
        try { h->qthread = std::thread(queue_thread_main, h); }
        catch (...) { return fail(h, TSDF_E_NOMEM, "tsdf_track_aos: cannot start the staging thread"); }
    }
    // the other set of pinned planes: the copies out of it were those of the frame before the last one
    std::swap(h->pin_xyz, h->alt_xyz); std::swap(h->pin_nrm, h->alt_nrm); std::swap(h->pin_rgb, h->alt_rgb);
    std::swap(h->ev_stage_done[0], h->ev_stage_done[1]); std::swap(h->stage_recorded[0], h->stage_recorded[1]);
    if (h->stage_recorded[0]) HIP_TRY(h, hipEventSynchronize(h->ev_stage_done[0]));
    lap(tp, h->sp.a_prep2);
    h->staged_xyz = false;
    h->tracked = tsdf_handle::TrackedCloud();
    if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);      // the frame this one replaces was never packed
    h->deferred = tsdf_handle::DeferredPack();
    // 1. the tracker's samples, straight from the cloud, in front of everything else (upload_samples_first)
    const int nb = h->fidx ^ 1;
    lap(tp, h->sp.a_prep);
    rc = upload_samples_first(h, points, (size_t)L->point_stride, (size_t)L->xyz_offset, width);
    if (rc) return rc;
    lap(tp, h->sp.a_gather);
    h->records_pending = false;              // (this frame's records are written by tsdf_integrate_aos, which orders them itself)
    choose_pixel_layout(h);
    h->fidx = nb; h->pn = h->pn_buf[nb]; h->samples = h->samples_buf[nb];
    h->have_frame = true;
    h->frame_serial++;
    h->frame_has_nrm = false;                // the pixel records are written when the normals arrive (tsdf_integrate_aos) ...
    h->frame_has_rgb = color;
    h->frame_side = true;
    // 2. the whole cloud -> pinned planes -> in_xyz / in_rgb, on the library threads and the frame stream, under the passes
    //    (tsdf_track_frame_aos: the normals as well -- one block, one copy -- and the pixel records behind them: the frame is
    //    complete when the passes are over and tsdf_integrate only waits for that packing on the device)
    {
        const tsdf_aos_layout lay = *L;
        PackArgs pa;
        if (normals) {
            rc = wait_buffer_free(h, nb, h->fstream);
            if (rc) return rc;
            pa = pack_args(h, h->in_xyz, h->in_nrm, color ? h->in_rgb : nullptr, h->pix_su, h->pix_sv, nb);
            pa.samples = nullptr;                // uploaded above
        }
        std::lock_guard<std::mutex> g(h->qmu);
        h->qbusy = true;
        h->queued.err = hipSuccess;
        h->qjob = [h, npix, points, normals, lay, color, pa] {
            float* const px = h->pin_xyz; float* const pnm = h->pin_nrm; uint8_t* const pc = h->pin_rgb;
            hipError_t e = stage_and_upload(h, npix, true, normals != nullptr, color, [&](size_t i0, size_t i1) {
                repack_aos(lay, points, normals, color, px, pnm, pc, i0, i1);
            }, normals ? 2 : 1);
            if (e == hipSuccess && normals) e = launch_pack(h->fstream, pa);
            if (e == hipSuccess && normals) e = hipEventRecord(h->ev_frame, h->fstream);
            h->queued.err = e;
        };
    }
    h->qcv.notify_all();
    lap(tp, h->sp.a_issue);
    // 3. estimate_new_position on the list
    const int rc_track = track_loop(h, stats);
    lap(tp, h->sp.a_loop);
    // 4. the cloud is the caller's again when this call returns
    {
        std::unique_lock<std::mutex> g(h->qmu);
        h->qcv.wait(g, [&] { return !h->qbusy; });
    }
    lap(tp, h->sp.a_wait);
    if (prof) h->sp.aos_frames++;
    if (h->queued.err != hipSuccess) return fail(h, TSDF_E_HIP, "tsdf_track_aos: staging the cloud failed: %s", hipGetErrorString(h->queued.err));
    HIP_TRY(h, hipEventRecord(h->ev_stage_done[0], h->fstream));      // the copies out of this set of planes, so far
    h->stage_recorded[0] = true;
    h->staged_xyz = true;
    h->tracked.valid = true; h->tracked.color = color; h->tracked.points = points; h->tracked.w = width; h->tracked.h = height;
    h->tracked.serial = h->frame_serial; h->tracked.lay = *L;
    h->tracked.normals = normals;
    if (normals) { h->frame_has_nrm = true; h->records_pending = true; }       // ... or are on their way already (ev_frame)
    return rc_track;
}
}  // namespace

int tsdf_integrate_aos(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L, int32_t width, int32_t height,
                       tsdf_integrate_stats* stats) {
    if (!h || !L || !normals || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_integrate_aos: bad argument (the normals are required)") : TSDF_E_BADARG;
    if (h->queued.active) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_integrate_aos");
    bool color = false;
    int rc = points ? check_point_layout(h, "tsdf_integrate_aos", L, &color) : TSDF_OK;
    if (rc) return rc;
    rc = check_normal_layout(h, "tsdf_integrate_aos", L);
    if (rc) return rc;
    using pclk = std::chrono::steady_clock;
    const bool prof = h->sp.on;
    auto lap = [prof](pclk::time_point& t, double& acc) { if (prof) { const pclk::time_point n = pclk::now(); acc += std::chrono::duration<double, std::nano>(n - t).count(); t = n; } };
    pclk::time_point tp = prof ? pclk::now() : pclk::time_point();
    {
        static const bool plain = [] { const char* e = std::getenv("TSDF_TRACK_AOS"); return e && std::atoi(e) == 0; }();
        if (plain) {                             // TSDF_TRACK_AOS=0 (comparison): round 4's shim -- the normals alone when a host frame of this size is current
            const bool keep = points && h->have_frame && h->staged_xyz && h->fw == width && h->fh == height && h->frame_serial > 1;
            rc = tsdf_set_frame_aos(h, keep ? nullptr : points, normals, L, width, height);
            return rc ? rc : tsdf_integrate(h, stats);
        }
    }
    if (h->tracked.valid && h->tracked.normals) {
        // The frame was tracked WITH its normals (tsdf_track_frame_aos) and is complete on the device.  This call still
        // integrates the clouds as they are NOW: both are compared with what was staged, every point; what changed (or
        // another cloud) goes up again.  A caller that vouches for its clouds calls tsdf_integrate instead and skips the
        // comparison (the shim's three-argument estimate_new_position does).
        const tsdf_handle::TrackedCloud& t = h->tracked;
        const bool ident = h->have_frame && h->staged_xyz && t.serial == h->frame_serial && t.w == width && t.h == height && normals == t.normals &&
                           L->normal_stride == t.lay.normal_stride && L->normal_offset == t.lay.normal_offset &&
                           (!points || (points == t.points && color == t.color && L->point_stride == t.lay.point_stride && L->xyz_offset == t.lay.xyz_offset &&
                                        L->r_offset == t.lay.r_offset && L->g_offset == t.lay.g_offset && L->b_offset == t.lay.b_offset));
        bool same = ident;
        if (ident) {
            const size_t npix = (size_t)width * height;
            const tsdf_aos_layout lay = *L;
            std::atomic<int> differs{0};
            const float* const px = h->pin_xyz; const float* const pnm = h->pin_nrm; const uint8_t* const pc = h->pin_rgb;
            const std::function<void(int, int)> verify = [&](int part, int parts) {
                const size_t i0 = npix * (size_t)part / (size_t)parts, i1 = npix * (size_t)(part + 1) / (size_t)parts;
                if ((points && !points_equal_planes(lay, points, color, px, pc, i0, i1)) || !normals_equal_plane(lay, normals, pnm, i0, i1))
                    differs.store(1, std::memory_order_relaxed);
            };
            HostPool* const pool = host_pool(h);
            if (pool) pool->run(verify); else verify(0, 1);
            same = differs.load() == 0;
        }
        h->tracked.valid = false;
        if (same) return tsdf_integrate(h, stats);
        rc = tsdf_set_frame_aos(h, points, normals, L, width, height);        // (points == NULL: the staged xyz / rgb are kept)
        return rc ? rc : tsdf_integrate(h, stats);
    }
    const tsdf_handle::TrackedCloud& tc = h->tracked;
    // is the frame in the library the cloud estimate_new_position was given?  Identity first (cheap), contents below.
    const bool candidate = tc.valid && h->have_frame && h->staged_xyz && tc.serial == h->frame_serial && tc.w == width && tc.h == height &&
                           (!points || (points == tc.points && color == tc.color && L->point_stride == tc.lay.point_stride && L->xyz_offset == tc.lay.xyz_offset &&
                                        L->r_offset == tc.lay.r_offset && L->g_offset == tc.lay.g_offset && L->b_offset == tc.lay.b_offset));
    if (!candidate) {
        // not the tracked cloud (or nothing was tracked through tsdf_track_aos): the whole frame goes up
        rc = tsdf_set_frame_aos(h, points, normals, L, width, height);
        if (rc) return rc;
        return tsdf_integrate(h, stats);
    }
    rc = bind_device(h);
    if (rc) return rc;
    if (!h->have_K) return fail(h, TSDF_E_NO_INTRINSICS, "camera matrix not received (reference: sdf.cpp:227-230 exits)");
    if (h->cfg.with_color && !h->frame_has_rgb) return fail(h, TSDF_E_NO_FRAME, "with_color=1 needs rgb in the current frame");
    // (Launching the integration's list_rows_kernel here, ahead of the normals -- the list needs the pose only -- was built
    // and measured in round 5: 8 alternations, median 2186 frames/s with it against 2311 without.  The launch call delays
    // the repack of the normals by as much as the kernel would later cost: profiles/r05_entry_points.json.)
    choose_pixel_layout(h);
    const size_t npix = (size_t)width * height;
    const tsdf_aos_layout lay = *L;
    HostPool* const pool = host_pool(h);
    float* const px = h->pin_xyz; float* const pnm = h->pin_nrm; uint8_t* const pc = h->pin_rgb;
    auto split = [npix](int part, int parts, size_t* i0, size_t* i1) {
        // multiples of four points, so that the 16-byte stores of the repack stay aligned in every part
        *i0 = (npix * (size_t)part / (size_t)parts) & ~(size_t)3; *i1 = part + 1 == parts ? npix : (npix * (size_t)(part + 1) / (size_t)parts) & ~(size_t)3;
    };
    // 1. the normals: repack into the pinned plane of the set that holds the cloud and copy (this copy is on the frame's
    //    critical path; TSDF_NORMAL_CHUNKS pieces repack piece c+1 while piece c travels -- measured, 8 alternations each:
    //    medians 2223 / 2186 / 2224 frames/s with 1 / 2 / 3 pieces, so one)
    {
        static const int kNc = [] { const char* e = std::getenv("TSDF_NORMAL_CHUNKS"); const int n = e ? std::atoi(e) : 1; return n < 1 ? 1 : n > 8 ? 8 : n; }();
        for (int c = 0; c < kNc; ++c) {
            const size_t c0 = (npix * (size_t)c / (size_t)kNc) & ~(size_t)3, c1 = c + 1 == kNc ? npix : (npix * (size_t)(c + 1) / (size_t)kNc) & ~(size_t)3;
            const std::function<void(int, int)> fill = [&](int part, int parts) {
                const size_t n = c1 - c0;
                const size_t i0 = c0 + ((n * (size_t)part / (size_t)parts) & ~(size_t)3), i1 = part + 1 == parts ? c1 : c0 + ((n * (size_t)(part + 1) / (size_t)parts) & ~(size_t)3);
                repack_aos(lay, nullptr, normals, false, nullptr, pnm, nullptr, i0, i1);
            };
            if (pool) pool->run(fill); else fill(0, 1);
            HIP_TRY(h, hipMemcpyAsync(h->in_nrm + 3 * c0, pnm + 3 * c0, (c1 - c0) * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
        }
    }
    lap(tp, h->sp.b_normals);
    // 2. under that copy: is `points` still, byte for byte, what was repacked when it was tracked?  (A cloud filtered in
    //    place between the two calls must be integrated as it is NOW: sdf.cpp:258-259 reads it at update time.)
    bool same = true;
    if (points) {
        std::atomic<int> differs{0};
        const std::function<void(int, int)> verify = [&](int part, int parts) {
            size_t i0, i1; split(part, parts, &i0, &i1);
            if (!points_equal_planes(lay, points, color, px, pc, i0, i1)) differs.store(1, std::memory_order_relaxed);
        };
        if (pool) pool->run(verify); else verify(0, 1);
        same = differs.load() == 0;
    }
    lap(tp, h->sp.b_verify);
    if (!same) {
        h->tracked.valid = false;
        HIP_TRY(h, stage_and_upload(h, npix, true, false, color, [&](size_t i0, size_t i1) {
            repack_aos(lay, points, nullptr, color, px, nullptr, pc, i0, i1);
        }));
    }
    HIP_TRY(h, hipEventRecord(h->ev_stage_done[0], h->fstream));
    h->stage_recorded[0] = true;
    // 3. the pixel records (and, for a changed cloud, its sample list), then SDF::update
    rc = wait_buffer_free(h, h->fidx, h->fstream);
    if (rc) return rc;
    {
        PackArgs pa = pack_args(h, h->in_xyz, h->in_nrm, h->frame_has_rgb ? h->in_rgb : nullptr, h->pix_su, h->pix_sv, h->fidx);
        if (same) pa.samples = nullptr;          // uploaded by tsdf_track_aos
        EventPair* ep;
        rc = timed_begin(h, 1, &ep, h->fstream);
        if (rc) return rc;
        HIP_TRY(h, launch_pack(h->fstream, pa));
        rc = timed_end(h, ep, h->fstream);
        if (rc) return rc;
    }
    HIP_TRY(h, hipEventRecord(h->ev_frame, h->fstream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_frame, 0));
    h->frame_has_nrm = true;
    h->frame_side = true;
    h->tracked.valid = false;                // one-shot: a second update of the same cloud uploads it
    lap(tp, h->sp.b_issue);
    rc = tsdf_integrate(h, stats);
    lap(tp, h->sp.b_integrate);
    return rc;
}

int tsdf_sample(tsdf_handle* h, const double* vox, int32_t n, float* val, int32_t* ok) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!vox || !val || !ok || n < 0) return fail(h, TSDF_E_BADARG, "tsdf_sample: bad argument");
    if (n == 0) return TSDF_OK;
    // Scratch kept in the handle and grown on demand: the reference's callers ask for one point at a time
    // (SDF::interpolate_distance), and a hipMalloc/hipFree pair per call would synchronise the whole device,
    // frame side stream included.
    if ((size_t)n > h->sample_cap) {
        if (h->sample_vox) (void)hipFree(h->sample_vox);
        if (h->sample_val) (void)hipFree(h->sample_val);
        if (h->sample_ok) (void)hipFree(h->sample_ok);
        h->sample_vox = nullptr; h->sample_val = nullptr; h->sample_ok = nullptr; h->sample_cap = 0;
        const size_t cap = (size_t)n < 256 ? 256 : (size_t)n;
        if (hipMalloc((void**)&h->sample_vox, cap * 3 * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&h->sample_val, cap * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&h->sample_ok, cap * sizeof(int32_t)) != hipSuccess) {
            (void)hipGetLastError();
            if (h->sample_vox) (void)hipFree(h->sample_vox);
            if (h->sample_val) (void)hipFree(h->sample_val);
            if (h->sample_ok) (void)hipFree(h->sample_ok);
            h->sample_vox = nullptr; h->sample_val = nullptr; h->sample_ok = nullptr;
            return fail(h, TSDF_E_NOMEM, "tsdf_sample: scratch for %d points", n);
        }
        h->sample_cap = cap;
    }
    hipError_t e = hipMemcpyAsync(h->sample_vox, vox, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = launch_sample(h->stream, h->grid, h->dw, h->sample_vox, n, h->sample_val, h->sample_ok);
    if (e == hipSuccess) e = hipMemcpyAsync(val, h->sample_val, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ok, h->sample_ok, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, TSDF_E_HIP, "tsdf_sample: %s", hipGetErrorString(e));
    for (int32_t i = 0; i < n; ++i)
        if (ok[i] < 0) return fail(h, TSDF_E_HALO, "tsdf_sample: point %d reads outside the stored layers", i);
    return TSDF_OK;
}

// ---- mesh extraction ---------------------------------------------------------------------------------

int tsdf_mesh_extract(tsdf_handle* h, float iso_level, int32_t with_color, int64_t* n_triangles) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (n_triangles) *n_triangles = 0;
    h->mesh_ntri = -1;
    if (!(iso_level >= 0.0f && iso_level < 1.0f))                 // marching_cubes_sdf.cpp:246-252
        return fail(h, TSDF_E_BADARG, "tsdf_mesh_extract: iso level %g outside [0,1)", (double)iso_level);
    if (with_color && !h->crgb) return fail(h, TSDF_E_BADARG, "tsdf_mesh_extract: the volume keeps no colour");
    const Grid& g = h->grid;
    MeshParams p{};
    p.g = g;
    p.extent[0] = h->cfg.width; p.extent[1] = h->cfg.height; p.extent[2] = h->cfg.depth;
    p.iso = iso_level;
    p.ci0 = g.own_x0 > 1 ? g.own_x0 : 1;
    p.ci1 = g.own_x1 < g.m - 1 ? g.own_x1 : g.m - 1;              // cube layers [ci0, ci1): base voxels 1..m-2
    if (p.ci1 > p.ci0 && g.xe < p.ci1 + 1)
        return fail(h, TSDF_E_HALO, "tsdf_mesh_extract: a sharded volume needs halo >= 1 (cube layer %d reads layer %d)",
                    p.ci1 - 1, p.ci1);
    const size_t n_rows = (size_t)mesh_rows(p);
    if (n_rows > (size_t)INT32_MAX) return fail(h, TSDF_E_BADARG, "tsdf_mesh_extract: too many rows");
    if (n_rows > h->mesh_rows_cap) {
        if (h->mesh_row_count) (void)hipFree(h->mesh_row_count);
        if (h->mesh_row_offset) (void)hipFree(h->mesh_row_offset);
        if (h->mesh_group_sum) (void)hipFree(h->mesh_group_sum);
        if (h->mesh_group_base) (void)hipFree(h->mesh_group_base);
        h->mesh_row_count = nullptr; h->mesh_row_offset = nullptr; h->mesh_rows_cap = 0;
        h->mesh_group_sum = nullptr; h->mesh_group_base = nullptr;
        const size_t n_groups = (size_t)mesh_scan_groups((long long)n_rows);
        if (hipMalloc((void**)&h->mesh_row_count, n_rows * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_row_offset, n_rows * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_group_sum, n_groups * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_group_base, n_groups * sizeof(unsigned long long)) != hipSuccess)
            return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: row tables (%zu rows)", n_rows);
        h->mesh_rows_cap = n_rows;
    }
    if (!h->mesh_total) HIP_TRY(h, hipHostMalloc((void**)&h->mesh_total, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    h->mesh_total[0] = 0ull; h->mesh_total[1] = 0ull;
    unsigned long long* d_total = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer((void**)&d_total, h->mesh_total, 0));
    HIP_TRY(h, launch_mesh_count(h->stream, p, h->dw, h->mesh_row_count, h->mesh_row_offset, h->mesh_group_sum,
                                 h->mesh_group_base, d_total));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const unsigned long long n = h->mesh_total[0];
    if (n > (unsigned long long)INT64_MAX / 64) return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: %llu triangles", n);
    if (n > h->mesh_verts_cap) {
        if (h->mesh_verts) (void)hipFree(h->mesh_verts);
        if (h->mesh_desc) (void)hipFree(h->mesh_desc);
        h->mesh_verts = nullptr; h->mesh_desc = nullptr; h->mesh_verts_cap = 0;
        const size_t cap = (size_t)n + (size_t)n / 8 + 1024;          // room to grow between calls
        if (hipMalloc((void**)&h->mesh_verts, cap * 9 * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_desc, cap * sizeof(unsigned long long)) != hipSuccess)
            return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: %llu triangles need %zu bytes", n, cap * 9 * sizeof(float));
        h->mesh_verts_cap = cap;
    }
    if (with_color && n > h->mesh_colors_cap) {
        if (h->mesh_colors) (void)hipFree(h->mesh_colors);
        h->mesh_colors = nullptr; h->mesh_colors_cap = 0;
        const size_t cap = h->mesh_verts_cap;
        if (hipMalloc((void**)&h->mesh_colors, cap * 3 * sizeof(float4)) != hipSuccess)
            return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: colours of %llu triangles", n);
        h->mesh_colors_cap = cap;
    }
    if (n) {
        HIP_TRY(h, launch_mesh_emit(h->stream, p, h->dw, h->crgb, h->mesh_row_count, h->mesh_row_offset, h->mesh_group_base,
                                    h->mesh_desc, h->mesh_verts,
                                    with_color ? h->mesh_colors : nullptr, n, (unsigned*)(d_total + 1)));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->mesh_total[1])
            return fail(h, TSDF_E_HALO, "tsdf_mesh_extract: a vertex colour reads outside the stored layers (halo >= 1 needed)");
    }
    h->mesh_ntri = (int64_t)n;
    h->mesh_has_color = with_color != 0;
    if (n_triangles) *n_triangles = (int64_t)n;
    return TSDF_OK;
}

int tsdf_mesh_read(tsdf_handle* h, float* vertices, float* colors, int64_t capacity_triangles) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (h->mesh_ntri < 0) return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: call tsdf_mesh_extract first");
    if (!vertices && h->mesh_ntri > 0) return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: vertices is NULL");
    if (capacity_triangles < h->mesh_ntri)
        return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: room for %lld triangles, the mesh has %lld",
                    (long long)capacity_triangles, (long long)h->mesh_ntri);
    if (colors && !h->mesh_has_color) return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: the last extraction had no colours");
    const size_t n = (size_t)h->mesh_ntri;
    if (n == 0) return TSDF_OK;
    HIP_TRY(h, hipMemcpyAsync(vertices, h->mesh_verts, n * 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (colors) HIP_TRY(h, hipMemcpyAsync(colors, h->mesh_colors, n * 12 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TSDF_OK;
}

int tsdf_mesh_device(tsdf_handle* h, const float** vertices, const float** colors, int64_t* n_triangles) {
    if (!h) return TSDF_E_BADARG;
    if (h->mesh_ntri < 0) return fail(h, TSDF_E_BADARG, "tsdf_mesh_device: call tsdf_mesh_extract first");
    if (vertices) *vertices = h->mesh_verts;
    if (colors) *colors = h->mesh_has_color ? reinterpret_cast<const float*>(h->mesh_colors) : nullptr;
    if (n_triangles) *n_triangles = h->mesh_ntri;
    return TSDF_OK;
}

// ---- volume I/O --------------------------------------------------------------------------------------

namespace {

// Copy `planes` float arrays of `n` voxels between host and the interleaved device layout, through a
// bounded device scratch (chunked so a 2048^3 slab does not need a second copy of itself).
int volume_io(tsdf_handle* h, bool download, bool color, int64_t first, int64_t n, float* const* host) {
    const int planes = color ? 4 : 2;
    const int64_t chunk = (int64_t)1 << 24;     // 16 Mi voxels per pass
    float* scratch = nullptr;
    const int64_t cap = n < chunk ? n : chunk;
    HIP_TRY(h, hipMalloc((void**)&scratch, (size_t)cap * planes * sizeof(float)));
    hipError_t e = hipSuccess;
    for (int64_t off = 0; off < n && e == hipSuccess; off += chunk) {
        const int64_t c = (n - off) < chunk ? (n - off) : chunk;
        float* pl[4] = {scratch, scratch + cap, scratch + 2 * cap, scratch + 3 * cap};
        if (download) {
            if (color) e = launch_split4(h->stream, h->crgb + first + off, pl[0], pl[1], pl[2], pl[3], c);
            else e = launch_split(h->stream, h->dw + first + off, pl[0], pl[1], c);
            for (int q = 0; q < planes && e == hipSuccess; ++q)
                e = hipMemcpyAsync(host[q] + off, pl[q], (size_t)c * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        } else {
            for (int q = 0; q < planes && e == hipSuccess; ++q)
                e = hipMemcpyAsync(pl[q], host[q] + off, (size_t)c * sizeof(float), hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) {
                if (color) e = launch_merge4(h->stream, h->crgb + first + off, pl[0], pl[1], pl[2], pl[3], c);
                else e = launch_merge(h->stream, h->dw + first + off, pl[0], pl[1], c);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(scratch);
    if (e != hipSuccess) return fail(h, TSDF_E_HIP, "volume I/O: %s", hipGetErrorString(e));
    return TSDF_OK;
}

}  // namespace

int tsdf_download(tsdf_handle* h, float* D, float* W) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!D || !W) return fail(h, TSDF_E_BADARG, "tsdf_download: null output");
    const int64_t mm = (int64_t)h->grid.m * h->grid.m;
    float* host[4] = {D, W, nullptr, nullptr};
    return volume_io(h, true, false, (h->grid.own_x0 - h->grid.xs) * mm, (h->grid.own_x1 - h->grid.own_x0) * mm, host);
}

int tsdf_upload(tsdf_handle* h, const float* D, const float* W) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!D || !W) return fail(h, TSDF_E_BADARG, "tsdf_upload: null input");
    const int64_t mm = (int64_t)h->grid.m * h->grid.m;
    float* host[4] = {const_cast<float*>(D), const_cast<float*>(W), nullptr, nullptr};
    return volume_io(h, false, false, (h->grid.own_x0 - h->grid.xs) * mm, (h->grid.own_x1 - h->grid.own_x0) * mm, host);
}

int tsdf_upload_with_halo(tsdf_handle* h, const float* D, const float* W) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!D || !W) return fail(h, TSDF_E_BADARG, "tsdf_upload_with_halo: null input");
    float* host[4] = {const_cast<float*>(D), const_cast<float*>(W), nullptr, nullptr};
    return volume_io(h, false, false, 0, h->n_stored, host);
}

int tsdf_download_color(tsdf_handle* h, float* Color_W, float* R, float* G, float* B) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!h->crgb) return fail(h, TSDF_E_BADARG, "volume was created with with_color=0");
    if (!Color_W || !R || !G || !B) return fail(h, TSDF_E_BADARG, "tsdf_download_color: null output");
    const int64_t mm = (int64_t)h->grid.m * h->grid.m;
    float* host[4] = {Color_W, R, G, B};
    return volume_io(h, true, true, (h->grid.own_x0 - h->grid.xs) * mm, (h->grid.own_x1 - h->grid.own_x0) * mm, host);
}

int tsdf_upload_color(tsdf_handle* h, const float* Color_W, const float* R, const float* G, const float* B) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!h->crgb) return fail(h, TSDF_E_BADARG, "volume was created with with_color=0");
    if (!Color_W || !R || !G || !B) return fail(h, TSDF_E_BADARG, "tsdf_upload_color: null input");
    const int64_t mm = (int64_t)h->grid.m * h->grid.m;
    float* host[4] = {const_cast<float*>(Color_W), const_cast<float*>(R), const_cast<float*>(G), const_cast<float*>(B)};
    return volume_io(h, false, true, (h->grid.own_x0 - h->grid.xs) * mm, (h->grid.own_x1 - h->grid.own_x0) * mm, host);
}

int tsdf_upload_color_with_halo(tsdf_handle* h, const float* Color_W, const float* R, const float* G, const float* B) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!h->crgb) return fail(h, TSDF_E_BADARG, "volume was created with with_color=0");
    if (!Color_W || !R || !G || !B) return fail(h, TSDF_E_BADARG, "tsdf_upload_color_with_halo: null input");
    float* host[4] = {const_cast<float*>(Color_W), const_cast<float*>(R), const_cast<float*>(G), const_cast<float*>(B)};
    return volume_io(h, false, true, 0, h->n_stored, host);
}

// ---- checkpoint ----------------------------------------------------------------------------------------

namespace {
struct VolHeader {                 // TSDFVOL2, little-endian, 80 bytes
    char magic[8];
    int32_t m, x0, x1, has_color;  // x0, x1: the slab the writer owned (informative)
    float width, height, depth, delta, epsilon;
    int32_t xs;                    // the file holds the x layers [xs, xe): the writer's slab AND its halo
    double origin[3];
    int32_t xe, reserved;
};
static_assert(sizeof(VolHeader) == 80, "checkpoint header layout");

bool read_plane(FILE* f, long long plane_floats, int plane, long long first, float* dst, size_t n) {
    const long long off = (long long)sizeof(VolHeader) + ((long long)plane * plane_floats + first) * (long long)sizeof(float);
    if (fseeko(f, (off_t)off, SEEK_SET) != 0) return false;
    return std::fread(dst, sizeof(float), n, f) == n;
}
}  // namespace

int tsdf_save(tsdf_handle* h, const char* path) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!path) return fail(h, TSDF_E_BADARG, "tsdf_save: null path");
    const size_t n = (size_t)h->n_stored;              // slab + halo: a restored shard needs its halo layers too
    std::vector<float> buf;
    try { buf.resize(n * 4); } catch (...) { return fail(h, TSDF_E_NOMEM, "tsdf_save: out of host memory"); }
    VolHeader hd;
    std::memset(&hd, 0, sizeof hd);
    std::memcpy(hd.magic, "TSDFVOL2", 8);
    hd.m = h->grid.m; hd.x0 = h->grid.own_x0; hd.x1 = h->grid.own_x1; hd.has_color = h->crgb ? 1 : 0;
    hd.xs = h->grid.xs; hd.xe = h->grid.xe;
    hd.width = h->cfg.width; hd.height = h->cfg.height; hd.depth = h->cfg.depth;
    hd.delta = h->cfg.delta; hd.epsilon = h->cfg.epsilon;
    std::memcpy(hd.origin, h->cfg.origin, sizeof hd.origin);
    // written next to the destination and renamed over it once complete: a failed save leaves neither a truncated file
    // with a valid header nor a destroyed earlier checkpoint
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(h, TSDF_E_BADARG, "tsdf_save: cannot open %s", tmp.c_str());
    bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1;
    float* host[4] = {buf.data(), buf.data() + n, buf.data() + 2 * n, buf.data() + 3 * n};
    if (ok) {
        rc = volume_io(h, true, false, 0, (int64_t)n, host);
        ok = rc == TSDF_OK && std::fwrite(buf.data(), sizeof(float), 2 * n, f) == 2 * n;
    }
    if (ok && h->crgb) {
        rc = volume_io(h, true, true, 0, (int64_t)n, host);
        ok = rc == TSDF_OK && std::fwrite(buf.data(), sizeof(float), 4 * n, f) == 4 * n;
    }
    ok = (std::fflush(f) == 0) && ok;
    ok = (std::fclose(f) == 0) && ok;
    if (rc || !ok) {
        std::remove(tmp.c_str());
        return rc ? rc : fail(h, TSDF_E_BADARG, "tsdf_save: short write to %s", tmp.c_str());
    }
    if (std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return fail(h, TSDF_E_BADARG, "tsdf_save: cannot rename %s to %s", tmp.c_str(), path);
    }
    return TSDF_OK;
}

int tsdf_load(tsdf_handle* h, const char* path) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!path) return fail(h, TSDF_E_BADARG, "tsdf_load: null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(h, TSDF_E_BADARG, "tsdf_load: cannot open %s", path);
    VolHeader hd;
    if (std::fread(&hd, sizeof hd, 1, f) != 1 || std::memcmp(hd.magic, "TSDFVOL2", 8) != 0) {
        std::fclose(f);
        return fail(h, TSDF_E_BADARG, "tsdf_load: %s is not a TSDFVOL2 file", path);
    }
    const Grid& g = h->grid;
    if (hd.m != g.m || (hd.has_color != 0) != (h->crgb != nullptr)) {
        std::fclose(f);
        return fail(h, TSDF_E_BADARG, "tsdf_load: file holds m=%d colour=%d, handle has m=%d colour=%d", hd.m, hd.has_color,
                    g.m, h->crgb ? 1 : 0);
    }
    // the geometry the voxel values were fused under must be the handle's (a D value means nothing under another delta)
    if (hd.width != h->cfg.width || hd.height != h->cfg.height || hd.depth != h->cfg.depth || hd.delta != h->cfg.delta ||
        hd.epsilon != h->cfg.epsilon || std::memcmp(hd.origin, h->cfg.origin, sizeof hd.origin) != 0) {
        std::fclose(f);
        return fail(h, TSDF_E_BADARG, "tsdf_load: %s was written for another volume (extent %gx%gx%g, origin %g %g %g, delta %g, epsilon %g)",
                    path, (double)hd.width, (double)hd.height, (double)hd.depth, hd.origin[0], hd.origin[1], hd.origin[2],
                    (double)hd.delta, (double)hd.epsilon);
    }
    // every STORED layer of this handle (slab and halo) must come from the file: a halo left at its old contents
    // would silently break the 'halo == neighbour's interior' invariant the sharded tracker relies on
    if (hd.xs > g.xs || hd.xe < g.xe || hd.xs < 0 || hd.xe > hd.m) {
        std::fclose(f);
        return fail(h, TSDF_E_HALO, "tsdf_load: file holds x layers [%d,%d), this handle stores [%d,%d) (slab [%d,%d) + halo %d)",
                    hd.xs, hd.xe, g.xs, g.xe, g.own_x0, g.own_x1, h->cfg.halo);
    }
    const size_t n = (size_t)h->n_stored;
    const long long mm = (long long)g.m * g.m;
    const long long plane_floats = (long long)(hd.xe - hd.xs) * mm, first = (long long)(g.xs - hd.xs) * mm;
    {
        // the whole file must be there BEFORE anything is uploaded: a file cut inside its colour part must not leave the
        // handle with new D / W and old colour
        const long long want = (long long)sizeof(VolHeader) + plane_floats * (long long)sizeof(float) * (hd.has_color ? 6 : 2);
        long long have = -1;
        if (fseeko(f, 0, SEEK_END) == 0) have = (long long)ftello(f);
        if (have < want || fseeko(f, (off_t)sizeof(VolHeader), SEEK_SET) != 0) {
            std::fclose(f);
            return fail(h, TSDF_E_BADARG, "tsdf_load: %s is truncated (%lld of %lld bytes); nothing was loaded", path, have, want);
        }
    }
    std::vector<float> buf;
    try { buf.resize(n * 4); } catch (...) { std::fclose(f); return fail(h, TSDF_E_NOMEM, "tsdf_load: out of host memory"); }
    float* host[4] = {buf.data(), buf.data() + n, buf.data() + 2 * n, buf.data() + 3 * n};
    bool ok = read_plane(f, plane_floats, 0, first, host[0], n) && read_plane(f, plane_floats, 1, first, host[1], n);
    if (ok) rc = volume_io(h, false, false, 0, (int64_t)n, host);
    if (ok && rc == TSDF_OK && h->crgb) {
        for (int q = 0; q < 4 && ok; ++q) ok = read_plane(f, plane_floats, 2 + q, first, host[q], n);
        if (ok) rc = volume_io(h, false, true, 0, (int64_t)n, host);
    }
    std::fclose(f);
    if (!ok) return fail(h, TSDF_E_BADARG, "tsdf_load: %s is truncated", path);
    return rc;
}

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

// Rendezvous on the named segment without help from the caller (all waits bounded by 20 s):
//   rank 0    removes whatever carries the name, creates the segment exclusively, zero-fills it, writes
//             {generation, nranks}, then the magic; waits until every other rank has written the generation into its
//             `joined` word; UNLINKS the name; then sets `go`.
//   rank r>0  opens the name (retrying while it does not exist or is still short), waits for the magic, writes the
//             generation it read into joined[r], waits for `go`; whenever the name turns out to designate another
//             object than the one mapped (a leftover of a crashed run that rank 0 has meanwhile replaced) it starts over.
// A segment whose `go` is set has no name any more, so a crashed run can never leave a segment behind that a later
// run could mistake for its own; what a crashed initialisation leaves has no `go`.  Every published word also carries the generation.
int tsdf_comm_init_shm(tsdf_handle* h, int32_t nranks, int32_t rank, const char* name) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!name || nranks <= 0 || nranks > 4096 || rank < 0 || rank >= nranks) return fail(h, TSDF_E_BADARG, "tsdf_comm_init_shm: bad argument");
    peer_close(h);
    shm_close(h);
    const size_t header = shm_header_bytes(nranks);
    const size_t bytes = header + (size_t)nranks * 2 * kShmSlot;
    const auto t0 = std::chrono::steady_clock::now();
    auto expired = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20); };
    auto nap = [] { struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr); };
    char* base = nullptr;
    unsigned long long gen = 0;
    if (rank == 0) {
        (void)shm_unlink(name);
        const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(h, TSDF_E_COMM, "shm_open(%s, O_EXCL) failed: %s", name, std::strerror(errno));
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return fail(h, TSDF_E_COMM, "ftruncate(%s) failed", name); }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { shm_unlink(name); return fail(h, TSDF_E_COMM, "mmap(%s) failed", name); }
        base = (char*)m;
        std::memset(base, 0, bytes);
        gen = ((unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 17)) & 0xFFFFFFFFull;
        if (gen == 0) gen = 1;
        *shm_hdr(base, kShmHdrGen) = gen;
        *shm_hdr(base, kShmHdrRanks) = (unsigned long long)nranks;
        __atomic_store_n(shm_hdr(base, kShmHdrJoined + 0), gen, __ATOMIC_RELAXED);
        __atomic_store_n(shm_hdr(base, kShmHdrMagic), kShmMagic, __ATOMIC_RELEASE);
        for (int r = 1; r < nranks; ++r) {
            while (__atomic_load_n(shm_hdr(base, kShmHdrJoined + r), __ATOMIC_ACQUIRE) != gen) {
                if (expired()) {
                    munmap(base, bytes); shm_unlink(name);
                    return fail(h, TSDF_E_COMM, "tsdf_comm_init_shm(%s): rank %d did not join within 20 s", name, r);
                }
                nap();
            }
        }
        shm_unlink(name);
        __atomic_store_n(shm_hdr(base, kShmHdrGo), gen, __ATOMIC_RELEASE);
    } else {
        for (;;) {
            if (expired()) return fail(h, TSDF_E_COMM, "tsdf_comm_init_shm(%s): rank 0's segment did not appear within 20 s", name);
            const int fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) { nap(); continue; }
            struct stat st;
            if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { close(fd); nap(); continue; }
            void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) return fail(h, TSDF_E_COMM, "mmap(%s) failed", name);
            base = (char*)m;
            // does the name still designate the object we mapped?  (false once rank 0 has replaced a leftover)
            auto replaced = [&] {
                struct stat now;
                const int f2 = shm_open(name, O_RDWR, 0600);
                if (f2 < 0) return false;              // no name: rank 0 unlinked it after the last join, `go` follows
                const bool other = fstat(f2, &now) == 0 && (now.st_ino != st.st_ino || now.st_dev != st.st_dev);
                close(f2);
                return other;
            };
            bool restart = false, joined = false;
            for (unsigned spins = 0;; ++spins) {
                if (!joined && __atomic_load_n(shm_hdr(base, kShmHdrMagic), __ATOMIC_ACQUIRE) == kShmMagic) {
                    if (*shm_hdr(base, kShmHdrRanks) != (unsigned long long)nranks) {
                        // a leftover of a crashed job of another size that rank 0 has not replaced yet: wait for the
                        // replacement (restart) and fail only if the name still designates this object when time is up
                        if (replaced()) { restart = true; break; }
                        if (expired()) {
                            munmap(base, bytes);
                            return fail(h, TSDF_E_COMM, "tsdf_comm_init_shm(%s): segment was made for another number of ranks", name);
                        }
                        nap();
                        continue;
                    }
                    gen = *shm_hdr(base, kShmHdrGen);
                    __atomic_store_n(shm_hdr(base, kShmHdrJoined + rank), gen, __ATOMIC_RELEASE);
                    joined = true;
                }
                if (joined && __atomic_load_n(shm_hdr(base, kShmHdrGo), __ATOMIC_ACQUIRE) == gen) break;
                if ((spins & 15u) == 15u) {
                    if (replaced()) { restart = true; break; }
                    if (expired()) { munmap(base, bytes); return fail(h, TSDF_E_COMM, "tsdf_comm_init_shm(%s): no go from rank 0 within 20 s", name); }
                }
                nap();
            }
            if (!restart) break;
            munmap(base, bytes);
            base = nullptr;
        }
    }
    // The device alias is only needed when a rank's final kernel writes its slot itself (host fold off); with the
    // default host fold the segment is touched by hosts only, so a failed registration is not fatal.
    hipError_t e = hipHostRegister(base, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    void* dptr = nullptr;
    if (e == hipSuccess) {
        e = hipHostGetDevicePointer(&dptr, base, 0);
        if (e != hipSuccess) { (void)hipHostUnregister(base); dptr = nullptr; }
    }
    if (e != hipSuccess) { (void)hipGetLastError(); dptr = nullptr; }
    h->shm.nranks = nranks; h->shm.rank = rank; h->shm.base = base; h->shm.dev_base = (char*)dptr;
    h->shm.bytes = bytes; h->shm.header = header; h->shm.gen = gen; h->shm.name = name;
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
        char* slot = h->shm.base + shm_slot_offset(h, h->shm.rank, seq);
        double row[kRedWidth];
        for (int e = 0; e < kRedWidth; ++e) row[e] = e < n ? buf[e] : 0.0;
        std::memcpy(slot, row, sizeof row);
        __atomic_store_n(reinterpret_cast<unsigned long long*>(slot + kRedWidth * sizeof(double)), shm_word(h, seq), __ATOMIC_RELEASE);
        int rc2 = shm_fan_in(h, seq, n);
        if (rc2) return rc2;
        std::memcpy(buf, h->red_host, (size_t)n * sizeof(double));
        return TSDF_OK;
    }
    if (h->hook) {
        if (h->hook(buf, n, h->hook_ctx) != 0) return fail(h, TSDF_E_COMM, "all-reduce hook reported failure");
    }
    return TSDF_OK;
}

// ---- host-only algebra ---------------------------------------------------------------------------------

int tsdf_host_set_pose(const double rot[9], const double trans[3], double rot_inv[9], double rot_inv_trans[3]) {
    if (!rot || !trans || !rot_inv || !rot_inv_trans) return TSDF_E_BADARG;
    hm::Pose P;
    hm::set_pose(P, rot, trans);
    std::memcpy(rot_inv, P.rot_inv, sizeof P.rot_inv);
    std::memcpy(rot_inv_trans, P.rot_inv_trans, sizeof P.rot_inv_trans);
    return TSDF_OK;
}

int tsdf_host_perturbed_rotations(const double rot[9], float w_h, double rpm[54]) {
    if (!rot || !rpm) return TSDF_E_BADARG;
    hm::Pose P;
    std::memcpy(P.rot, rot, sizeof P.rot);
    hm::perturbed_rotations(P, w_h, rpm);
    return TSDF_OK;
}

int tsdf_host_gn_step(double rot[9], double trans[3], const double A[36], const double b[6],
                      float max_twist_diff, double twist[6], int32_t* stop) {
    if (!rot || !trans || !A || !b) return TSDF_E_BADARG;
    hm::Pose P;
    hm::set_pose(P, rot, trans);
    double tw[6];
    bool st = false;
    if (!hm::gn_step(P, A, b, max_twist_diff, tw, &st)) return TSDF_E_SINGULAR;
    std::memcpy(rot, P.rot, sizeof P.rot);
    std::memcpy(trans, P.trans, sizeof P.trans);
    if (twist) std::memcpy(twist, tw, sizeof tw);
    if (stop) *stop = st ? 1 : 0;
    return TSDF_OK;
}

// ---- measurement -------------------------------------------------------------------------------------

int tsdf_set_timing(tsdf_handle* h, int32_t on) {
    if (!h) return TSDF_E_BADARG;
    int rc = bind_device(h);
    if (rc) return rc;
    if (!on) { rc = drain_events(h); if (rc) return rc; }
    h->timing = (on & 1) != 0;
    h->timing_track = (on & 2) != 0;
    h->timing_period = (on >> 8) & 0xFF;
    if (h->timing_period < 1) h->timing_period = 1;
    h->timing_seen[0] = h->timing_seen[1] = 0u;
    return TSDF_OK;
}

int tsdf_read_timing(tsdf_handle* h, tsdf_timing* out, int32_t reset) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    rc = drain_events(h);
    if (rc) return rc;
    if (out) *out = h->tm;
    if (reset) h->tm = tsdf_timing{};
    return TSDF_OK;
}

int tsdf_read_counters(tsdf_handle* h, tsdf_counters* out, int32_t reset) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    rc = fetch_counters(h);
    if (rc) return rc;
    h->cnt.n_updated = (int64_t)(h->counters_host[kCntUpdatedOwned] - h->cnt_base[kCntUpdatedOwned]);
    h->cnt.n_updated_halo = (int64_t)(h->counters_host[kCntUpdatedHalo] - h->cnt_base[kCntUpdatedHalo]);
    h->cnt.integrate_items = (int64_t)(h->counters_host[kCntItems] - h->cnt_base[kCntItems]);
    if (out) *out = h->cnt;
    if (reset) {
        for (int i = 0; i < kNumCounters; ++i) h->cnt_base[i] = h->counters_host[i];
        h->cnt = tsdf_counters{};
    }
    return TSDF_OK;
}

int tsdf_synchronize(tsdf_handle* h) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    {   // a queued pageable frame: its copies and pack are only on the frame stream once the staging thread has issued them
        std::unique_lock<std::mutex> g(h->qmu);
        h->qcv.wait(g, [&] { return !h->qbusy; });
    }
    // Device frames whose packing is still deferred: tsdf_synchronize ends the library's claim on borrowed device
    // planes (tsdf.h), so what has not been packed yet is packed now, by a launch of its own
    if (h->deferred.pending) {
        PackArgs own = pack_args(h, h->deferred.xyz, h->deferred.nrm, h->deferred.rgb, h->pix_su, h->pix_sv, h->fidx);
        if (h->deferred.samples_listed) own.samples = nullptr;
        HIP_TRY(h, launch_pack(h->stream, own));
        h->deferred.pending = false;
    }
    if (h->queued.active && h->queued.device && h->queued.deferred && !h->queued.packed) {
        tsdf_handle::Queued& q = h->queued;
        pick_pixel_layout(h, &q.su, &q.sv);
        HIP_TRY(h, launch_pack(h->stream, pack_args(h, q.d_xyz, q.d_nrm, q.d_rgb, q.su, q.sv, q.nb)));
        q.packed = true;
    }
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->aql.wait_idle();
    h->borrowed.clear();                     // nothing launched so far reads a borrowed plane any more
    h->borrow_lost = -1;
    return TSDF_OK;
}

// Serial (as tsdf_frame_serial counts) of the newest frame such that the library no longer reads the DEVICE planes of
// that frame or of any frame before it.  Never blocks, launches nothing.
int64_t tsdf_device_frame_released(const tsdf_handle* h) {
    if (!h) return -1;
    return released_serial(const_cast<tsdf_handle*>(h));      // (drops the entries that have become free: bookkeeping only)
}

void* tsdf_stream(tsdf_handle* h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"
