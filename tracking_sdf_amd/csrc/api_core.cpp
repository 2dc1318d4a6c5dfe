// api_core.cpp -- handle life cycle, camera state, timing, counters, synchronisation (see handle.hpp).
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

#include <dlfcn.h>

namespace {
std::mutex g_err_mu;
std::string g_create_error;
}  // namespace

namespace tsdf_api {

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

int bind_device(tsdf_handle* h) {
    HIP_TRY(h, hipSetDevice(h->device));
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

// Cumulative device counters into h->counters_host (synchronises the main stream): the item count comes from the
// counter block, the updated-voxel counts are the sums of the integrate workgroups' own words.
int fetch_counters(tsdf_handle* h) {
    const size_t nw = 2 * (size_t)h->integrate_blocks;
    HIP_TRY(h, hipMemcpyAsync(h->counters_host, h->counters, kNumCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->wg_counts_host, h->wg_counts, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned long long own = 0, halo = 0;
    for (size_t b = 0; b < nw; b += 2) { own += h->wg_counts_host[b]; halo += h->wg_counts_host[b + 1]; }
    h->counters_host[kCntUpdatedOwned] = own;
    h->counters_host[kCntUpdatedHalo] = halo;
    return TSDF_OK;
}

int check_ready(tsdf_handle* h, bool need_frame) {
    if (!h) return TSDF_E_BADARG;
    if (need_frame && !h->have_frame) return fail(h, TSDF_E_NO_FRAME, "no frame: call tsdf_set_frame first");
    return bind_device(h);
}

}  // namespace tsdf_api

// =================================================================================================
extern "C" {

int tsdf_abi_version(void) { return TSDF_ABI_VERSION; }

const char* tsdf_last_error(const tsdf_handle* h) {
    if (h) return h->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_create_error.c_str();
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
    const int32_t stride = cfg->slab_stride;
    if (stride < 0) return fail(nullptr, TSDF_E_BADARG, "tsdf_create: slab_stride %d", stride);
    if (stride > 0) {
        const int32_t B = x1 - x0, Lb = B + 2 * cfg->halo;
        if ((cfg->m & (cfg->m - 1)) != 0 || cfg->m % B != 0 || x0 % B != 0 || stride % B != 0 || stride < Lb)
            return fail(nullptr, TSDF_E_BADARG, "tsdf_create: block-cyclic placement needs m a power of two and a multiple of the block (%d), "
                        "blocks at multiples of it, and stride (%d) >= block + 2 * halo (%d)", B, stride, Lb);
        const unsigned long long layers = (unsigned long long)((cfg->m - x0 + stride - 1) / stride) * (unsigned long long)Lb;
        if (layers * Lb * Lb >= (1ull << 32))                              // (the multiply-high division of grid_global_layer)
            return fail(nullptr, TSDF_E_BADARG, "tsdf_create: %llu stored layers in blocks of %d are too many for the block arithmetic", layers, Lb);
    }

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
    if (stride > 0) {
        g.blk_own = x1 - x0; g.blk_layers = g.blk_own + 2 * cfg->halo; g.blk_stride = stride; g.blk_first = x0 - cfg->halo;
        g.n_blocks = (cfg->m - x0 + stride - 1) / stride;
        g.blk_magic = (uint32_t)((1ull << 32) / (unsigned)g.blk_layers + 1ull);
    }
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
    // (Stream priorities -- main stream highest, frame stream lowest -- and a CU mask on the frame stream were measured in
    // round 6: no gain / a loss for every host-frame path, profiles/r06_frame_stream_policies.json.)
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&h->fstream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_frame, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_samples, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_stage_done[0], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_stage_done[1], hipEventDisableTiming));
    h->n_stored = (int64_t)grid_stored_layers(g) * g.m * g.m;
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
        // TSDF_INTEGRATE_BLOCKS_PER_CU: persistent workgroups per CU at most (default: what the occupancy query allows)
        const char* env = std::getenv("TSDF_INTEGRATE_BLOCKS_PER_CU");
        const int per_cu = env ? std::atoi(env) : integrate_blocks_per_cu();
        h->integrate_blocks = (prop.multiProcessorCount * (per_cu > 0 ? per_cu : 4) + 7) / 8 * 8;   // whole XCD groups
        h->integrate_cus = prop.multiProcessorCount;
    }
    // 2 words per workgroup (updated voxels: owned, halo)
    CREATE_TRY(hipMalloc((void**)&h->wg_counts, 2 * (size_t)h->integrate_blocks * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(h->wg_counts, 0, 2 * (size_t)h->integrate_blocks * sizeof(unsigned long long), h->stream));
    CREATE_TRY(hipHostMalloc((void**)&h->wg_counts_host, 2 * (size_t)h->integrate_blocks * sizeof(unsigned long long), hipHostMallocDefault));
    CREATE_TRY(hipMalloc((void**)&h->red_dev, kRedWidth * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&h->red_host, (kRedWidth + 2) * sizeof(double), hipHostMallocDefault));
    std::memset(h->red_host, 0, (kRedWidth + 2) * sizeof(double));
    CREATE_TRY(hipHostMalloc((void**)&h->release_host, 4 * sizeof(unsigned long long), hipHostMallocDefault));
    h->release_host[0] = h->release_host[1] = 0ull;
    h->release_host[2] = ~0ull;              // work items of the last integrate launch (none yet: the full grid)
    { const char* ev = std::getenv("TSDF_DEFER_PACK"); h->defer_device_pack = !(ev && std::atoi(ev) == 0); h->deferred_list_samples = !(ev && std::atoi(ev) == 2); }
    { const char* ev = std::getenv("TSDF_HOST_FOLD"); h->host_fold = !(ev && std::atoi(ev) == 0); }
    CREATE_TRY(hipHostMalloc((void**)&h->shard_host, (size_t)kTrackShards * kShardSlotDoubles * sizeof(double), hipHostMallocDefault));
    std::memset(h->shard_host, 0, (size_t)kTrackShards * kShardSlotDoubles * sizeof(double));
    { const char* ev = std::getenv("TSDF_PROFILE"); const int bits = ev ? std::atoi(ev) : 0; h->sp.on = (bits & 1) != 0; h->track_profile = (bits & 2) != 0; }
    CREATE_TRY(hipMalloc((void**)&h->fold_ctr, 2 * track_fold_counter_words() * sizeof(unsigned)));
    CREATE_TRY(hipMemsetAsync(h->fold_ctr, 0, 2 * track_fold_counter_words() * sizeof(unsigned), h->stream));
    {
        const char* ev = std::getenv("TSDF_AQL");
        if (ev && std::atoi(ev) != 0) {
            // the code object sits next to this library
            Dl_info info;
            std::string path;
            if (dladdr(reinterpret_cast<const void*>(&tsdf_abi_version), &info) && info.dli_fname) {
                path = info.dli_fname;
                const size_t slash = path.find_last_of('/');
                std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
                const std::string dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
                // lib<name>_hip.so -> <name>_track.hsaco (the Makefile's rule); any other library name: <file name>_track.hsaco
                const bool plain = base.size() > 10 && base.compare(0, 3, "lib") == 0 && base.compare(base.size() - 7, 7, "_hip.so") == 0;
                base = (plain ? base.substr(3, base.size() - 10) : base) + "_track.hsaco";
                path = dir + "/" + base;
            }
            std::string why;
            h->aql_on = !path.empty() && h->aql.init(h->device, path.c_str(), track_kernel_symbol_prefix(), track_kernel_explicit_arg_bytes(), track_kernel_build_id(), &why);
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
        wait_staging_job(h, 0);
        { std::lock_guard<std::mutex> g(h->qmu); h->qstop = true; }
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
    }
    if (h->ev_frame) (void)hipEventDestroy(h->ev_frame);
    if (h->ev_copied) (void)hipEventDestroy(h->ev_copied);
    if (h->ev_order) (void)hipEventDestroy(h->ev_order);
    if (h->ev_samples) (void)hipEventDestroy(h->ev_samples);
    for (int b = 0; b < tsdf_handle::kQueueBlocks; ++b) if (h->ev_qblk[b]) (void)hipEventDestroy(h->ev_qblk[b]);
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
    wait_staging_job(h, 0);    // queued pageable frames: their copies are only on the frame stream once the staging thread has issued them
    // Device frames whose packing is still deferred: tsdf_synchronize ends the library's claim on borrowed device
    // planes (tsdf.h), so what has not been packed yet is packed now, by a launch of its own
    if (h->deferred.pending) {
        if (h->records_pending) {             // a samples-first host frame: its planes may still be travelling on the frame stream
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_frame, 0));
            h->records_pending = false;
        }
        PackArgs own = pack_args(h, h->deferred.xyz, h->deferred.nrm, h->deferred.rgb, h->pix_su, h->pix_sv, h->fidx);
        if (h->deferred.samples_listed) own.samples = nullptr;
        HIP_TRY(h, launch_pack(h->stream, own));
        h->deferred.pending = false;
    }
    if (h->qcount > 0 && h->queued_front().device && !h->queued_front().packed) {      // (deferred or, TSDF_DEFER_PACK=0, waiting for tsdf_next_frame)
        tsdf_handle::Queued& q = h->queued_front();
        pick_pixel_layout(h, &q.su, &q.sv);
        HIP_TRY(h, launch_pack(h->stream, pack_args(h, q.d_xyz, q.d_nrm, q.d_rgb, q.su, q.sv, h->fidx ^ 1)));
        q.packed = true;
    }
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->aql_on && !h->aql.wait_idle()) {
        h->aql_on = false;
        return fail(h, TSDF_E_HIP, "tsdf_synchronize: the library's own queue did not become idle (queue disabled; borrowed frames stay borrowed)");
    }
    h->samples_written_ticket[0] = h->samples_written_ticket[1] = 0ull;      // ... or writes a sample list
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
