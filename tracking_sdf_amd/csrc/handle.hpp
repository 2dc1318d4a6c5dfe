// handle.hpp -- the handle behind the C ABI of include/tsdf.h and what the api_*.cpp translation units share.
//   api_core.cpp     create / destroy / reset, pose and intrinsics, timing, counters, synchronisation
//   api_frames.cpp   frames: staging of host buffers, the frame queue, device frames, depth pre-processing
//   api_hotpath.cpp  the Gauss-Newton loop (reference src/camera_tracking.cpp:66-245) and the integrate launch
//                    (src/sdf.cpp:224-315), the reference's two calls on its own clouds, tsdf_sample
//   api_comm.cpp     ranks: RCCL, shared-memory fan-in, device-side peer exchange
//   api_volume.cpp   host mirrors, checkpoints, mesh extraction
//   host_util.cpp    everything that needs no device (also built under sanitizers)
// No torch types, no exceptions across the boundary, no CPU fallback.
#pragma once

#include "../../include/tsdf.h"

#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "aql_queue.hpp"
#include "host_math.hpp"
#include "host_util.hpp"
#include "rccl_dyn.hpp"
#include "tsdf_device.h"

namespace tsdf_api {
struct EventPair { hipEvent_t a = nullptr, b = nullptr; };
using tsdf::host::HostPool;
}  // namespace tsdf_api

using namespace tsdf;      // (the members of the handle are tsdf:: types; this header is included by the api_*.cpp files only)

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
    // Frame queue (tsdf_queue_frame* / tsdf_next_frame): up to kQueueDepth frames wait behind the current one while it is
    // tracked and integrated -- their planes travel to (or already sit in) device memory meanwhile.  A frame handed over
    // in DEVICE memory only as the first of them (its records are packed ahead, into the record buffer the current frame
    // does not use); host and depth frames in either place: the second one gives the staging + copy of a frame two frame
    // times instead of one (the copy alone is 155 us of a 190 us frame: profiles/r06_host_queue.json).
    static constexpr int kQueueDepth = 2;
    struct Queued {
        bool active = false, direct = false, device = false, has_nrm = false, has_rgb = false;
        int blk = -1;                      // >= 0: a host / depth frame whose planes are (being) put into qblk[blk]
        // device frames with deferred packing: nothing is launched when the frame is queued; the current frame's integrate
        // launch packs it (packed = true), or it becomes current unpacked like a frame of tsdf_set_frame_device
        bool deferred = false, packed = false;
        const float* d_xyz = nullptr; const float* d_nrm = nullptr; const uint8_t* d_rgb = nullptr;
        int32_t su = 1, sv = 0;
        uint64_t job = 0;                  // > 0: the staging thread's job for this frame (done once qdone >= job)
        hipError_t err = hipSuccess;       // what the staging thread's HIP calls returned
        int rc = 0;                        // tsdf_queue_depth_frame: what the pre-processing on the staging thread returned
        std::string msg;                   // ... and its message, handed to the handle by tsdf_next_frame
    } qslot[kQueueDepth];
    int qhead = 0, qcount = 0;             // the queue: qslot[qhead] is the frame tsdf_next_frame takes next
    Queued& queued_front() { return qslot[qhead]; }
    Queued& queued_slot(int pos) { return qslot[(qhead + pos) % kQueueDepth]; }
    std::thread qthread;                   // runs the pageable path's staging so that the caller can go on tracking
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<std::function<void()>> qjobs;   // in order, one at a time (they share the pinned staging sets and the frame stream's order)
    uint64_t qissued = 0, qdone = 0;       // jobs handed to the thread / finished (under qmu)
    bool qstop = false;
    hipError_t stage_err = hipSuccess;     // tsdf_track_aos: what its staging job's HIP calls returned
    void* worklist = nullptr;      // integrate work items (32-byte descriptors: row << 6 | chunk, the row's share of rot_inv * g)
    unsigned* work_count = nullptr;   // work-list bookkeeping (two alternating sets: item count, band histogram, cursors)
    int integrate_blocks = 0;      // persistent grid of integrate_kernel: the most workgroups a launch uses (CUs x workgroups per CU)
    int integrate_cus = 0;         // CUs (rounded up to whole XCD groups of 8 workgroups)

    // frame
    int32_t fw = 0, fh = 0, ncols = 0, nrows = 0, n_samples = 0;
    int32_t pix_su = 1, pix_sv = 0;   // layout of the packed pixel records of the current frame
    bool have_frame = false, frame_has_nrm = false, frame_has_rgb = false;
    float* pin_xyz = nullptr; float* pin_nrm = nullptr; uint8_t* pin_rgb = nullptr;  // pinned host staging (the set in use)
    // second staging set of the frame queue's pageable path: frame k+1 is filled into one set while the DMA engine still
    // reads frame k from the other (with one set the caller's thread waited for those copies before every queue call)
    float* alt_xyz = nullptr; float* alt_nrm = nullptr; uint8_t* alt_rgb = nullptr; size_t alt_cap = 0;
    // tsdf_track_aos / tsdf_integrate_aos (the reference's two calls on its own clouds): the tracker's samples go up first,
    // through their own pinned list; the cloud estimate_new_position was called with, for SDF::update's "same cloud?" check
    float4* pin_samples[2] = {nullptr, nullptr}; size_t pin_samples_cap = 0;   // [0]: of the staging set in use, [1]: of the other (swapped with them:
                                                                               // a set's ev_stage_done then covers the list's copy as well)
    struct TrackedCloud {
        bool valid = false, color = false;
        const void* points = nullptr; int32_t w = 0, h = 0; int64_t serial = -1;
        const void* normals = nullptr;         // non-null: tsdf_track_frame_aos staged the normals too (the frame is complete)
        tsdf_aos_layout lay{};
    } tracked;
    hipEvent_t ev_stage_done[2] = {nullptr, nullptr};   // [0]: the copies out of the set in use have been issued up to here; [1]: the other set's
    bool stage_recorded[2] = {false, false};
    size_t in_cap = 0;             // pixels the staging buffers hold
    bool staged_xyz = false;       // the library holds the planes of the CURRENT frame on the device (host / AoS / depth frames) ...
    const float* staged_planes[2] = {nullptr, nullptr};   // ... here: xyz, nrm of a block of the ring below
    int staged_blk = -1;           // the ring block that holds them
    // Frames that come through the QUEUE from host memory or as raw depth (round 6): their planes land in one of kQueueBlocks
    // device blocks (xyz | nrm | rgb, the layout of the pinned staging sets) and are packed like a frame handed over in device memory --
    // by workgroups appended to the frame's OWN integrate launch, the first tracker pass reading its samples from the xyz
    // plane -- instead of by a pack_kernel of their own on the frame stream, which ran next to the current frame's
    // latency-bound tracker passes (profiles/r06_host_queue.json: 4270-4470 -> 4900+ frames/s).  A block is reused once
    // the launch that packed its frame has run (the release tickets of tsdf_device_frame_released).
    static constexpr int kQueueBlocks = kQueueDepth + 2;   // the current frame's, the queued frames', and one whose integrate launch may still run
    char* qblk[kQueueBlocks] = {};
    size_t qblk_cap = 0;                                   // pixels a block holds
    int64_t qblk_serial[kQueueBlocks] = {};         // serial of the frame whose planes the block holds unpacked (0: none)
    hipEvent_t ev_qblk[kQueueBlocks] = {};   // frame stream: the block's planes are complete
    int qblk_next = 0;
    std::unique_ptr<tsdf_api::HostPool> pool;   // staging threads, started by the first pageable frame (TSDF_HOST_THREADS, default: usable cores - 2, at most 12)
    float* pre_z = nullptr; float* pre_zf = nullptr; void* pre_depth = nullptr; void* pin_depth = nullptr;   // pre-processing scratch
    size_t pre_cap = 0;
    float2* pre_grid_a = nullptr; float2* pre_grid_b = nullptr; size_t pre_grid_cap = 0;                     // bilateral grid (cells)
    unsigned* pre_minmax = nullptr; unsigned* pin_minmax = nullptr;                                           // depth range words
    float4* pn = nullptr;          // 2 x float4 per pixel      } the CURRENT frame's buffers: one of the two below
    float4* samples = nullptr;     //                            }
    size_t pn_cap = 0, samples_cap = 0;
    // Frame side stream: the H2D copies and the depth pre-processing of a frame run on `fstream`, so that they overlap
    // whatever the main stream is doing (the previous frame's integration, this frame's tracker passes).  Every pack --
    // inside an integrate launch, or a pack_kernel of its own with TSDF_DEFER_PACK=0 -- runs on the MAIN stream since round 6:
    // the record buffers need no cross-stream ordering any more.  `ev_frame`: the planes of the current frame are complete.
    hipStream_t fstream = nullptr;
    hipEvent_t ev_frame = nullptr;
    hipEvent_t ev_samples = nullptr;           // the frame's sample list is on the device (samples-first uploads)
    bool records_pending = false;              // the frame's planes are still being produced on the frame stream (ev_frame): the tracker
                                               // may run (it reads the sample list, sent ahead), tsdf_integrate waits for them
    hipEvent_t ev_copied = nullptr;            // the H2D copies of a frame handed over in page-locked caller buffers
    // A sample list that travels on the frame stream ("samples first") lands in samples_buf[nb] unordered against the main
    // stream, where the launch that packed an EARLIER frame into the same buffer (two frames ago: deferred / fused packing
    // writes the list too) may not have run yet when the host is several untracked frames ahead of the GPU -- it would
    // then overwrite the newer list.  samples_written_ticket[nb]: main-stream release ticket of the last such launch
    // (0: none outstanding); the copy is issued at once when the ticket has appeared in release_host[0] (always, in a
    // tracked stream), else behind ev_order recorded on the main stream.
    unsigned long long samples_written_ticket[2] = {0ull, 0ull};
    hipEvent_t ev_order = nullptr;
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
        bool internal;                         // the planes are a block of the library's own ring (a host / depth frame): not the caller's business
    };
    std::deque<BorrowedFrame> borrowed;
    int64_t borrow_lost = -1;                      // >= 0: an entry could not be recorded for this serial (see borrow_device_frame)
    unsigned long long* release_host = nullptr;    // pinned: [0], [1] the two streams' tickets; [2] work items of the last integrate launch
    unsigned long long release_ticket[2] = {0ull, 0ull};
    bool deferred_list_samples = true;         // TSDF_DEFER_PACK=2 (diagnosis): every pass reads the plane
    bool defer_device_pack = true;             // TSDF_DEFER_PACK=0: pack when the frame is set
    float4* pn_buf[2] = {nullptr, nullptr};
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
    double* shard_host = nullptr;  // pinned: kTrackShards slots of kShardSlotDoubles (host side of the fan-in)
    unsigned integrate_launches = 0;

    // TSDF_PROFILE bit 1: host-side clock of a pass, printed by tsdf_destroy (ns sums: parameters, launch call, wait
    // for the row, fold + solve + pose)
    bool track_profile = false;
    double tp_fill = 0, tp_launch = 0, tp_wait = 0, tp_post = 0; long long tp_passes = 0;
    // TSDF_PROFILE bit 0: host-side clock of the pageable-frame staging (ns sums per staged frame, printed by tsdf_destroy)
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
    // shared-memory fan-in (ranks of one node): nranks x 2 slots (double-buffered by pass parity) -- host_util.hpp
    tsdf::host::ShmSegment shm;

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
    std::vector<tsdf_api::EventPair> ev_pool;   // pending integrate/pack pairs
    std::vector<int> ev_kind;         // 0 = integrate, 1 = pack
    size_t ev_used = 0;
    tsdf_api::EventPair ev_track;
    tsdf_timing tm{};
    tsdf_counters cnt{};
    unsigned long long cnt_base[kNumCounters]{};

    std::string err;
};

namespace tsdf_api {
using namespace tsdf;

static_assert(kRedWidth == host::kShmRowDoubles, "the shared segment's rows are tracker result rows");

// ---- api_core.cpp
int fail(tsdf_handle* h, int code, const char* fmt, ...);
// Where fail() leaves its message when it runs on the queue's library thread (tsdf_queue_depth_frame): the handle's
// own string belongs to the caller's thread, which may be failing a call of its own at that moment.
extern thread_local std::string* t_err_sink;
int bind_device(tsdf_handle* h);
int check_ready(tsdf_handle* h, bool need_frame);
int drain_events(tsdf_handle* h);
int timed_begin(tsdf_handle* h, int kind, EventPair** out, hipStream_t st);
int timed_end(tsdf_handle* h, EventPair* ep, hipStream_t st);
int fetch_counters(tsdf_handle* h);
constexpr size_t kVolumePadFront = 16;   // voxels of {0,0} padding in front of the volume (128 bytes)

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return tsdf_api::fail((h), TSDF_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

// ---- api_frames.cpp
struct DevPlanes { float* xyz = nullptr; float* nrm = nullptr; uint8_t* rgb = nullptr; };   // where a staged frame is copied to
void free_preproc(tsdf_handle* h);
void free_frame(tsdf_handle* h);
int ensure_frame_buffers(tsdf_handle* h, int32_t w, int32_t hh, bool need_staging);
void pick_pixel_layout(const tsdf_handle* h, int32_t* su, int32_t* sv);
void choose_pixel_layout(tsdf_handle* h);
PackArgs pack_args(const tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t su, int32_t sv, int nb);
void borrow_device_frame(tsdf_handle* h, int64_t serial, bool internal = false);
ReleaseWord release_for(tsdf_handle* h, int64_t serial, int s);
void abandon_device_frame(tsdf_handle* h, int64_t serial);
int64_t released_serial(tsdf_handle* h, bool own_blocks_too = false);
int run_pack(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, bool registered);
int ensure_pin_samples(tsdf_handle* h);
int upload_samples_first(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width);
int stage_samples(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width, int nb);   // ... without the main stream's wait
bool samples_first_enabled();
int defer_pack(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, bool already_borrowed = false,
               bool own_block = false /* the planes are a block of the library's own ring */);
HostPool* host_pool(tsdf_handle* h);        // the staging threads, started by the first pageable frame
// work of a frame's staging job in front of its chunks: every pool thread does its share, the caller issues what they made
struct StageFirst {
    std::function<void(int, int)> work;      // (part, parts)
    std::function<hipError_t()> issue;
};
hipError_t stage_and_upload(tsdf_handle* h, size_t npix, bool has_xyz, bool has_nrm, bool has_rgb,
                            const std::function<void(size_t, size_t)>& fill, int chunks_when_unset, const struct DevPlanes* dst,
                            const StageFirst* first = nullptr);
StageFirst samples_first_work(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width, int nb, bool main_stream_waits);
int next_staging_set(tsdf_handle* h, size_t npix);
int staging_set_copies_issued(tsdf_handle* h);
int ensure_second_staging_set(tsdf_handle* h, size_t npix);
int acquire_queue_block(tsdf_handle* h, int* blk, DevPlanes* planes);      // a free block of the ring of device blocks
DevPlanes block_planes(const tsdf_handle* h, int blk);
int block_frame_current(tsdf_handle* h, int blk, const DevPlanes& p, bool has_nrm, bool has_rgb, bool samples_listed, bool travelling);
void queue_thread_main(tsdf_handle* h);
// hands a job to the staging thread (started on first use); 0: the thread could not be started
uint64_t submit_staging_job(tsdf_handle* h, std::function<void()> job);
// until that job has finished (0: until the thread is idle)
void wait_staging_job(tsdf_handle* h, uint64_t job);

// ---- api_hotpath.cpp
int accumulate_pass(tsdf_handle* h, bool reduce_ranks, bool later_pass = false /* pass >= 1 of a tsdf_track call */);
int track_loop(tsdf_handle* h, tsdf_track_stats* stats);

// ---- api_comm.cpp
void peer_close(tsdf_handle* h);
void shm_close(tsdf_handle* h);
constexpr int kPeerTimeoutMs = 5000;
PeerExchange peer_exchange_for(const tsdf_handle* h, unsigned long long seq);

}  // namespace tsdf_api
