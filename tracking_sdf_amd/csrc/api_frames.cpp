// api_frames.cpp -- how a frame reaches the device: staging of host buffers, the frame queue, device frames with
// deferred packing, depth pre-processing (see handle.hpp).
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

namespace tsdf_api {

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


void free_frame(tsdf_handle* h) {
    // (xyz | nrm | rgb live in ONE block each: the xyz pointer is the block)
    if (h->pin_xyz) (void)hipHostFree(h->pin_xyz);
    if (h->alt_xyz) (void)hipHostFree(h->alt_xyz);
    for (int b = 0; b < 2; ++b) { if (h->pin_samples[b]) (void)hipHostFree(h->pin_samples[b]); h->pin_samples[b] = nullptr; }
    h->pin_samples_cap = 0;
    h->tracked = tsdf_handle::TrackedCloud();
    for (int b = 0; b < tsdf_handle::kQueueBlocks; ++b) { if (h->qblk[b]) (void)hipFree(h->qblk[b]); h->qblk[b] = nullptr; h->qblk_serial[b] = 0; }
    h->qblk_cap = 0;
    h->staged_planes[0] = h->staged_planes[1] = nullptr;
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
    }
    if (npix > h->pn_cap) {
        for (int b = 0; b < 2; ++b) { if (h->pn_buf[b]) (void)hipFree(h->pn_buf[b]); h->pn_buf[b] = nullptr; }
        h->pn = nullptr; h->pn_cap = 0; h->have_frame = false;
        for (int b = 0; b < 2; ++b) HIP_TRY(h, hipMalloc((void**)&h->pn_buf[b], npix * kPixelRecordBytes));
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
        char* pin = nullptr;
        HIP_TRY(h, hipHostMalloc((void**)&pin, frame_block_bytes(npix), hipHostMallocDefault));
        h->pin_xyz = reinterpret_cast<float*>(pin); h->pin_nrm = reinterpret_cast<float*>(pin + plane); h->pin_rgb = reinterpret_cast<uint8_t*>(pin + 2 * plane);
        h->in_cap = npix;
    }
    h->fw = w; h->fh = hh; h->ncols = ncols; h->nrows = nrows; h->n_samples = (int32_t)ns;
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

PackArgs pack_args(const tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t su, int32_t sv, int nb) {
    PackArgs a;
    a.xyz = xyz; a.nrm = nrm; a.rgb = rgb;
    a.width = h->fw; a.height = h->fh; a.stride = h->cfg.pixel_stride;
    a.pix_su = su; a.pix_sv = sv;
    a.pn = h->pn_buf[nb]; a.samples = h->samples_buf[nb];
    a.ncols = h->ncols; a.nrows = h->nrows;
    a.color_layout = h->cfg.with_color ? 1 : 0;
    return a;
}

// ---- borrowed device planes (tsdf_device_frame_released) ---------------------------------------------------------
// a device frame with this serial has been handed over; nothing has packed it yet
void borrow_device_frame(tsdf_handle* h, int64_t serial, bool internal) {
    try { h->borrowed.push_back({serial, -1, 0ull, internal}); }
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
// newest serial S such that no device frame with serial <= S is still read by the library.  The caller's view
// (tsdf_device_frame_released) counts the planes the CALLER handed over in device memory; own_blocks_too also counts the
// blocks of the library's own ring, which hold host / depth frames until their integrate launch has packed them.
int64_t released_serial(tsdf_handle* h, bool own_blocks_too) {
    auto pending = [h](const tsdf_handle::BorrowedFrame& b) {
        return b.stream < 0 || (b.ticket && __atomic_load_n(h->release_host + b.stream, __ATOMIC_ACQUIRE) < b.ticket);
    };
    while (!h->borrowed.empty() && !pending(h->borrowed.front())) h->borrowed.pop_front();
    int64_t rel = h->frame_serial + ((h->qcount > 0 && h->queued_front().device) ? 1 : 0);   // (a device frame waits at the front only)
    for (const auto& b : h->borrowed)
        if ((own_blocks_too || !b.internal) && pending(b)) { rel = b.serial - 1; break; }
    if (h->borrow_lost >= 0 && rel >= h->borrow_lost) rel = h->borrow_lost - 1;
    return rel;
}

// TSDF_DEFER_PACK=0 only (rounds 1-3's form, kept for same-box comparisons): a frame handed over in DEVICE memory is packed
// at once, by a pack_kernel launch of its own on the main stream.  (Until round 5 host and depth frames were packed like
// this on the frame stream; they now go through the ring of device blocks and deferred packing, like everything else.)
// registered: the frame's entry in the list of borrowed frames exists already (a queued frame).
int run_pack(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, bool registered) {
    if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);      // the frame this one replaces was never packed
    choose_pixel_layout(h);
    const int nb = h->fidx ^ 1;                               // the buffer the previous frame did not use
    if (h->qcount > 0) return fail(h, TSDF_E_BADARG, "a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first");
    EventPair* ep;
    int rc = timed_begin(h, 1, &ep, h->stream);
    if (rc) return rc;
    HIP_TRY(h, launch_pack(h->stream, pack_args(h, xyz, nrm, rgb, h->pix_su, h->pix_sv, nb)));
    if (!registered) borrow_device_frame(h, h->frame_serial + 1);
    const ReleaseWord packed = release_for(h, h->frame_serial + 1, 0);
    HIP_TRY(h, launch_release(h->stream, packed));
    h->samples_written_ticket[nb] = packed.ticket;
    rc = timed_end(h, ep, h->stream);
    if (rc) return rc;
    h->records_pending = false;
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
bool samples_first_enabled() {
    static const bool on = [] { const char* e = std::getenv("TSDF_SAMPLES_FIRST"); return !(e && std::atoi(e) == 0); }();
    return on;
}
int upload_samples_first(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width) {
    int rc = ensure_pin_samples(h);
    if (rc) return rc;
    rc = stage_samples(h, base, pixel_bytes, xyz_offset, width, h->fidx ^ 1);
    if (rc) return rc;
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_samples, 0));
    return TSDF_OK;
}
// gather + copy + ev_samples on the frame stream (pin_samples must exist: ensure_pin_samples on the caller's thread)
int stage_samples(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width, int nb) {
    const StageFirst f = samples_first_work(h, base, pixel_bytes, xyz_offset, width, nb, false);
    HostPool* const pool = host_pool(h);
    if (pool) pool->run(f.work); else f.work(0, 1);
    HIP_TRY(h, f.issue());
    return TSDF_OK;
}
// The same as part of a frame's staging job (stage_and_upload's `first`): the workers gather their shares of the sample
// rows before they fill their shares of the planes, the caller issues the list's copy in front of the planes' -- and, with
// `main_stream_waits`, makes the main stream wait for that copy alone (upload_samples_first's rule).
StageFirst samples_first_work(tsdf_handle* h, const void* base, size_t pixel_bytes, size_t xyz_offset, int32_t width, int nb, bool main_stream_waits) {
    float4* const ps = h->pin_samples[0];       // the list buffer of the staging set in use (they swap together: next_staging_set)
    const int32_t st = h->cfg.pixel_stride, ncols = h->ncols, nrows = h->nrows;
    StageFirst f;
    f.work = [=](int part, int parts) {
        const int r0 = (int)((long long)nrows * part / parts), r1 = (int)((long long)nrows * (part + 1) / parts);
        gather_samples(base, pixel_bytes, xyz_offset, width, st, ncols, nrows, r0, r1, reinterpret_cast<float*>(ps));
    };
    f.issue = [=]() -> hipError_t {
        hipError_t e = hipSuccess;
        // (see samples_written_ticket: an earlier launch on the main stream may still have this buffer's list to write)
        const unsigned long long need = h->samples_written_ticket[nb];
        if (need && __atomic_load_n(h->release_host + 0, __ATOMIC_ACQUIRE) < need) {
            e = hipEventRecord(h->ev_order, h->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->fstream, h->ev_order, 0);
        }
        h->samples_written_ticket[nb] = 0ull;
        if (e == hipSuccess) e = hipMemcpyAsync(h->samples_buf[nb], ps, (size_t)h->n_samples * sizeof(float4), hipMemcpyHostToDevice, h->fstream);
        if (e == hipSuccess) e = hipEventRecord(h->ev_samples, h->fstream);
        if (e == hipSuccess && main_stream_waits) e = hipStreamWaitEvent(h->stream, h->ev_samples, 0);
        return e;
    };
    return f;
}

// The other set of pinned staging planes, free of the copies that last read it (those of the frame before the last one):
// a frame handed over one at a time no longer waits for the PREVIOUS frame's copy before it may fill its planes
// (hipStreamSynchronize(fstream) at the top of tsdf_set_frame until round 6: ~40 us of every frame).
int next_staging_set(tsdf_handle* h, size_t npix) {
    const int rc = ensure_second_staging_set(h, npix);
    if (rc) return rc;
    std::swap(h->pin_xyz, h->alt_xyz); std::swap(h->pin_nrm, h->alt_nrm); std::swap(h->pin_rgb, h->alt_rgb); std::swap(h->pin_samples[0], h->pin_samples[1]);
    std::swap(h->ev_stage_done[0], h->ev_stage_done[1]); std::swap(h->stage_recorded[0], h->stage_recorded[1]);
    if (h->stage_recorded[0]) HIP_TRY(h, hipEventSynchronize(h->ev_stage_done[0]));
    return TSDF_OK;
}
int staging_set_copies_issued(tsdf_handle* h) {
    HIP_TRY(h, hipEventRecord(h->ev_stage_done[0], h->fstream));
    h->stage_recorded[0] = true;
    return TSDF_OK;
}

// A frame handed over in device memory is not packed when it is set: the tracker reads its samples from the xyz plane
// (TrackParams::xyz_plane) and the pixel records are written inside the integrate launch, by workgroups appended to
// list_rows_kernel (launch_integrate) -- the packing then hides under that kernel's latency chain instead of being 11 us
// of its own in front of the first tracker pass.  The planes stay borrowed until that launch has run: tsdf.h asks for them
// until tsdf_device_frame_released() reaches the frame's serial (or tsdf_synchronize, which packs what is pending).
// TSDF_DEFER_PACK=0: pack at once, as rounds 1-3 did.
int defer_pack(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, bool already_borrowed, bool own_block) {
    if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);      // the frame this one replaces was never packed
    if (!already_borrowed) borrow_device_frame(h, h->frame_serial + 1, own_block);
    choose_pixel_layout(h);
    const int nb = h->fidx ^ 1;
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

}  // namespace tsdf_api

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
}  // namespace

namespace tsdf_api {
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
// `first` (optional): work of the same pool job in front of the chunks -- every worker does its share (first->work), the
// caller issues what they made (first->issue) before the first chunk's copy: the tracker's sample list of "samples first",
// which until round 6 was a pool job of its own (one more wake-up of eleven sleeping threads, 20-50 us, in front of every frame).
hipError_t stage_and_upload(tsdf_handle* h, size_t npix, bool has_xyz, bool has_nrm, bool has_rgb,
                            const std::function<void(size_t, size_t)>& fill, int chunks_when_unset, const DevPlanes* dst,
                            const StageFirst* first) {
    constexpr int kMaxChunks = 16;
    float* const d_xyz = dst->xyz; float* const d_nrm = dst->nrm; uint8_t* const d_rgb = dst->rgb;
    // TSDF_STAGE_CHUNKS overrides; otherwise the caller's choice: 1 where only throughput counts (the frame queue), 2 where
    // the frame's LATENCY to the device is on the critical path (tsdf_track_frame_aos: medians 2570 / 2850 / 2790 / 2760
    // frames/s with 1 / 2 / 3 / 4 pieces, six alternations)
    static const int kEnvChunks = [] { const char* e = std::getenv("TSDF_STAGE_CHUNKS"); const int n = e ? std::atoi(e) : 0; return n < 0 ? 0 : n > kMaxChunks ? kMaxChunks : n; }();
    const int kChunks = kEnvChunks > 0 ? kEnvChunks : (chunks_when_unset < 1 ? 1 : chunks_when_unset > kMaxChunks ? kMaxChunks : chunks_when_unset);
    std::atomic<int> done[kMaxChunks];
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<int> first_done{0};
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
            err = hipMemcpyAsync(d_xyz, h->pin_xyz, bytes, hipMemcpyHostToDevice, h->fstream);
            return;
        }
        if (has_xyz) err = hipMemcpyAsync(d_xyz + 3 * i0, h->pin_xyz + 3 * i0, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream);
        if (has_nrm && err == hipSuccess) err = hipMemcpyAsync(d_nrm + 3 * i0, h->pin_nrm + 3 * i0, n * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream);
        if (has_rgb && err == hipSuccess) err = hipMemcpyAsync(d_rgb + 3 * i0, h->pin_rgb + 3 * i0, n * 3, hipMemcpyHostToDevice, h->fstream);
    };
    HostPool* const pool = host_pool(h);
    const std::function<void(int, int)> job = [&](int part, int parts) {
        if (parts == 1) {                                   // no workers: fill and issue in turn (the DMA still overlaps)
            if (first) { first->work(0, 1); err = first->issue(); }
            for (int c = 0; c < kChunks; ++c) { fill(chunk_lo(c), chunk_lo(c + 1)); upload(c); }
        } else if (part == 0) {                             // the caller: HIP calls only
            if (first) {
                while (first_done.load(std::memory_order_acquire) < parts - 1) std::this_thread::yield();
                err = first->issue();
            }
            for (int c = 0; c < kChunks; ++c) {
                while (done[c].load(std::memory_order_acquire) < parts - 1) std::this_thread::yield();
                upload(c);
            }
        } else {
            const size_t wk = (size_t)(part - 1), nw = (size_t)(parts - 1);
            if (first) { first->work(part - 1, parts - 1); first_done.fetch_add(1, std::memory_order_release); }
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
}  // namespace tsdf_api

namespace tsdf_api {
DevPlanes block_planes(const tsdf_handle* h, int blk) {
    const size_t plane = plane_stride_bytes(h->in_cap);
    DevPlanes p;
    p.xyz = reinterpret_cast<float*>(h->qblk[blk]);
    p.nrm = reinterpret_cast<float*>(h->qblk[blk] + plane);
    p.rgb = reinterpret_cast<uint8_t*>(h->qblk[blk] + 2 * plane);
    return p;
}

// A block of the ring of device blocks for a frame that arrives from host memory (or as raw depth): allocated on first use (xyz | nrm | rgb,
// the layout of the pinned staging sets), free of the frame it held before -- that frame's planes were packed by its own integrate launch, which
// publishes a release ticket (tsdf_device_frame_released's mechanism); two frames later it has long run, so the wait below
// is a formality, bounded and backed by a real synchronisation.
int acquire_queue_block(tsdf_handle* h, int* blk, DevPlanes* planes) {
    if (h->qblk_cap != h->in_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->fstream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (int b = 0; b < tsdf_handle::kQueueBlocks; ++b) { if (h->qblk[b]) (void)hipFree(h->qblk[b]); h->qblk[b] = nullptr; h->qblk_serial[b] = 0; }
        h->qblk_cap = 0;
        for (int b = 0; b < tsdf_handle::kQueueBlocks; ++b) {
            HIP_TRY(h, hipMalloc((void**)&h->qblk[b], frame_block_bytes(h->in_cap)));
            if (!h->ev_qblk[b]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_qblk[b], hipEventDisableTiming));
        }
        h->qblk_cap = h->in_cap;
    }
    const int b = h->qblk_next;
    h->qblk_next = (b + 1) % tsdf_handle::kQueueBlocks;
    if (h->qblk_serial[b]) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0; released_serial(h, true) < h->qblk_serial[b]; ++spins)
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));      // (a frame that was never integrated keeps its block until here)
                break;
            }
        h->qblk_serial[b] = 0;
    }
    *planes = block_planes(h, b);
    *blk = b;
    return TSDF_OK;
}

// The frame whose planes are (being) written into ring block `blk` becomes the current frame with its packing deferred
// to its own integrate launch (defer_pack).  samples_listed: the tracker's sample list went up ahead (upload_samples_first),
// the passes do not touch the planes.  travelling: the planes are still being produced on the frame stream (ev_frame is
// recorded behind them here): with the samples listed only tsdf_integrate waits for them (records_pending), otherwise the
// main stream does at once -- the first tracker pass reads its samples from the xyz plane.
int block_frame_current(tsdf_handle* h, int blk, const DevPlanes& p, bool has_nrm, bool has_rgb, bool samples_listed, bool travelling) {
    if (travelling) HIP_TRY(h, hipEventRecord(h->ev_frame, h->fstream));
    const int rc = defer_pack(h, p.xyz, has_nrm ? p.nrm : nullptr, has_rgb ? p.rgb : nullptr, false, true);
    if (rc) return rc;
    h->deferred.samples_listed = samples_listed;
    if (travelling) {
        if (samples_listed) h->records_pending = true;
        else HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_frame, 0));
    }
    h->qblk_serial[blk] = h->frame_serial;               // the block is this frame's until the launch that packs it has run
    h->staged_xyz = true; h->staged_planes[0] = p.xyz; h->staged_planes[1] = p.nrm; h->staged_blk = blk;
    return TSDF_OK;
}
}  // namespace tsdf_api

int tsdf_set_frame(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t width, int32_t height) {
    if (!h || !xyz || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_set_frame: bad argument") : TSDF_E_BADARG;
    if (h->qcount > 0) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_set_frame");
    int rc = bind_device(h);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    // Page-locked caller buffers are copied from directly (no staging pass through the library's own pinned buffers:
    // at 640x480 that memcpy is 8.3 MB per frame, longer than the frame's GPU work); the copies are complete when the
    // call returns, so the buffers are borrowed for the call only, as for pageable ones.
    const bool direct = is_pinned_host(xyz, npix * 12) && (!nrm || is_pinned_host(nrm, npix * 12)) && (!rgb || is_pinned_host(rgb, npix * 3));
    h->staged_xyz = false;                                 // until this frame's planes are complete on the device
    // The planes go into a block of the ring and are packed by the frame's own integrate launch (as a frame handed over in
    // device memory is; round 6: no pack_kernel on the frame stream).
    int blk = -1;
    DevPlanes dst;
    rc = acquire_queue_block(h, &blk, &dst);
    if (rc) return rc;
    if (!direct) {
        // the staging set the frame before the last one used (its copies are long done; the previous frame's may still run)
        rc = next_staging_set(h, npix);
        if (rc) return rc;
        const bool samples_first = samples_first_enabled();
        StageFirst first;
        if (samples_first) {
            rc = ensure_pin_samples(h);
            if (rc) return rc;
            first = samples_first_work(h, xyz, 12, 0, width, h->fidx ^ 1, true);
        }
        HIP_TRY(h, stage_and_upload(h, npix, true, nrm != nullptr, rgb != nullptr, [&](size_t i0, size_t i1) {
            std::memcpy(h->pin_xyz + 3 * i0, xyz + 3 * i0, (i1 - i0) * 3 * sizeof(float));
            if (nrm) std::memcpy(h->pin_nrm + 3 * i0, nrm + 3 * i0, (i1 - i0) * 3 * sizeof(float));
            if (rgb) std::memcpy(h->pin_rgb + 3 * i0, rgb + 3 * i0, (i1 - i0) * 3);
        }, 1, &dst, samples_first ? &first : nullptr));
        rc = staging_set_copies_issued(h);
        if (rc) return rc;
        return block_frame_current(h, blk, dst, nrm != nullptr, rgb != nullptr, samples_first, true);
    }
    HIP_TRY(h, hipMemcpyAsync(dst.xyz, xyz, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
    if (nrm) HIP_TRY(h, hipMemcpyAsync(dst.nrm, nrm, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
    if (rgb) HIP_TRY(h, hipMemcpyAsync(dst.rgb, rgb, npix * 3, hipMemcpyHostToDevice, h->fstream));
    HIP_TRY(h, hipEventRecord(h->ev_copied, h->fstream));
    HIP_TRY(h, hipEventSynchronize(h->ev_copied));      // the caller's buffers have been read -- and the planes are complete
    return block_frame_current(h, blk, dst, nrm != nullptr, rgb != nullptr, false, false);
}


// ---- two-deep frame queue ----------------------------------------------------------------------------------------
namespace tsdf_api {
void queue_thread_main(tsdf_handle* h) {
    (void)hipSetDevice(h->device);
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> g(h->qmu);
            h->qcv.wait(g, [&] { return h->qstop || !h->qjobs.empty(); });
            if (h->qjobs.empty()) return;                   // (qstop is only set once the thread is idle)
            job.swap(h->qjobs.front());
            h->qjobs.pop_front();
        }
        job();
        { std::lock_guard<std::mutex> g(h->qmu); h->qdone++; }
        h->qcv.notify_all();
    }
}

uint64_t submit_staging_job(tsdf_handle* h, std::function<void()> job) {
    if (!h->qthread.joinable()) {
        try { h->qthread = std::thread(queue_thread_main, h); }
        catch (...) { return 0; }
    }
    uint64_t id = 0;
    try {
        std::lock_guard<std::mutex> g(h->qmu);
        h->qjobs.push_back(std::move(job));
        id = ++h->qissued;
    } catch (...) { return 0; }
    h->qcv.notify_all();
    return id;
}

void wait_staging_job(tsdf_handle* h, uint64_t job) {
    std::unique_lock<std::mutex> g(h->qmu);
    h->qcv.wait(g, [&] { return h->qdone >= (job ? job : h->qissued); });
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
}  // namespace tsdf_api

namespace {
// May one more frame be queued?  Behind an empty queue anything; behind one queued frame a host or depth frame (`from_host`),
// of the same size as everything else in flight.
int queue_admit(tsdf_handle* h, bool from_host, int32_t width, int32_t height) {
    if (h->qcount >= tsdf_handle::kQueueDepth)
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: %d frames are queued already (the queue holds the current frame + %d)", h->qcount, tsdf_handle::kQueueDepth);
    if (h->qcount > 0 && !from_host)
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame_device: a frame is queued already (a frame in device memory is taken as the first queued frame only)");
    if ((h->have_frame || h->qcount > 0) && (h->fw != width || h->fh != height))
        return fail(h, TSDF_E_BADARG, "tsdf_queue_frame: the queued frame must have the size of the current one (%dx%d)", h->fw, h->fh);
    return TSDF_OK;
}

// what both queue entry points share.  `fill` is null for page-locked plane buffers (copied from directly).
int queue_frame_common(tsdf_handle* h, const float* xyz, const float* nrm, const uint8_t* rgb, int32_t width, int32_t height,
                       bool has_nrm, bool has_rgb, std::function<void(size_t, size_t)> fill) {
    int rc = bind_device(h);
    if (rc) return rc;
    rc = queue_admit(h, true, width, height);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, true);
    if (rc) return rc;
    const size_t npix = (size_t)width * height;
    tsdf_handle::Queued& q = h->queued_slot(h->qcount);
    q = tsdf_handle::Queued();
    q.has_nrm = has_nrm; q.has_rgb = has_rgb; q.direct = !fill;
    // The frame's planes go into a block of the queue's ring and stay there, unpacked, until the frame is current: its own
    // integrate launch packs them (tsdf_next_frame -> defer_pack).  Nothing but the copy runs on the frame stream.
    DevPlanes dst;
    rc = acquire_queue_block(h, &q.blk, &dst);
    if (rc) return rc;
    const int blk = q.blk;
    if (q.direct) {
        HIP_TRY(h, hipMemcpyAsync(dst.xyz, xyz, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
        if (nrm) HIP_TRY(h, hipMemcpyAsync(dst.nrm, nrm, npix * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
        if (rgb) HIP_TRY(h, hipMemcpyAsync(dst.rgb, rgb, npix * 3, hipMemcpyHostToDevice, h->fstream));
        HIP_TRY(h, hipEventRecord(h->ev_qblk[blk], h->fstream));
        q.active = true; h->qcount++;
        return TSDF_OK;
    }
    // pageable buffers: a library thread fills pinned staging planes (with the staging pool) and issues copies and pack
    // while the caller goes on.  Two sets of staging planes alternate: the one that is filled now last fed the copies of
    // the frame before the current one, and the library thread, not the caller, waits for those if it has to.
    rc = ensure_second_staging_set(h, npix);
    if (rc) return rc;
    // (No "samples first" here, unlike frames handed over one at a time: a queued frame's copy runs under the frame before
    // it, and gathering its sample list on the library thread cost more than the passes' shorter wait gave back --
    // profiles/r06_host_queue.json, "samples_first_in_queue".)
    const auto t_queued = std::chrono::steady_clock::now();
    tsdf_handle::Queued* const slot = &q;
    q.job = submit_staging_job(h, [h, npix, has_nrm, has_rgb, fill, dst, blk, t_queued, slot] {
            const auto ts0 = std::chrono::steady_clock::now();
            if (h->sp.on) h->sp.handoff += std::chrono::duration<double, std::nano>(ts0 - t_queued).count();
            // switch to the other staging set (fill and stage_and_upload read h->pin_* when they run)
            std::swap(h->pin_xyz, h->alt_xyz); std::swap(h->pin_nrm, h->alt_nrm); std::swap(h->pin_rgb, h->alt_rgb); std::swap(h->pin_samples[0], h->pin_samples[1]);
            std::swap(h->ev_stage_done[0], h->ev_stage_done[1]); std::swap(h->stage_recorded[0], h->stage_recorded[1]);
            hipError_t e = h->stage_recorded[0] ? hipEventSynchronize(h->ev_stage_done[0]) : hipSuccess;
            if (h->sp.on) h->sp.sync_before += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - ts0).count();
            if (e == hipSuccess) e = stage_and_upload(h, npix, true, has_nrm, has_rgb, fill, 1, &dst, nullptr);
            if (e == hipSuccess) { e = hipEventRecord(h->ev_stage_done[0], h->fstream); h->stage_recorded[0] = e == hipSuccess; }
            if (e == hipSuccess) e = hipEventRecord(h->ev_qblk[blk], h->fstream);
            slot->err = e;
        });
    if (!q.job) return fail(h, TSDF_E_NOMEM, "tsdf_queue_frame: cannot start the staging thread");
    q.active = true; h->qcount++;
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
    rc = queue_admit(h, false, width, height);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, false);
    if (rc) return rc;
    tsdf_handle::Queued& q = h->queued_slot(0);
    q = tsdf_handle::Queued();
    q.has_nrm = d_nrm != nullptr; q.has_rgb = d_rgb != nullptr; q.direct = true; q.device = true;
    // No launch now.  Default: the integrate launch of the CURRENT frame packs this one in workgroups appended to its
    // list_rows_kernel (tsdf_integrate), on the main stream, i.e. behind the last reader of the record buffer (q.deferred).
    // TSDF_DEFER_PACK=0: a pack_kernel launch of its own when the frame becomes current (tsdf_next_frame).
    q.deferred = h->defer_device_pack; q.packed = false;
    q.d_xyz = d_xyz; q.d_nrm = d_nrm; q.d_rgb = d_rgb;
    q.active = true; h->qcount = 1;
    borrow_device_frame(h, h->frame_serial + 1);
    return TSDF_OK;
}

int tsdf_next_frame(tsdf_handle* h) {
    if (!h) return TSDF_E_BADARG;
    if (h->qcount == 0) return fail(h, TSDF_E_NO_FRAME, "tsdf_next_frame: no frame is queued");
    int rc = bind_device(h);
    if (rc) return rc;
    tsdf_handle::Queued& q = h->queued_front();
    h->qhead = (h->qhead + 1) % tsdf_handle::kQueueDepth; h->qcount--;     // (q stays where it is: the slot is rewritten by the next tsdf_queue_frame*)
    q.active = false;
    const bool from_device = q.device;
    q.device = false;
    if (from_device) {
        const bool deferred = q.deferred;
        q.deferred = false;
        h->staged_xyz = false;
        if (!q.packed)         // no launch has packed it yet: as tsdf_set_frame_device would (deferred, or a launch of its own: TSDF_DEFER_PACK=0)
            return deferred ? defer_pack(h, q.d_xyz, q.d_nrm, q.d_rgb, true) : run_pack(h, q.d_xyz, q.d_nrm, q.d_rgb, true);
        // packed inside the previous frame's integrate launch (or by tsdf_synchronize), on the main stream: nothing to wait for
        if (h->deferred.pending) abandon_device_frame(h, h->frame_serial);
        const int nb = h->fidx ^ 1;                        // where the launch that packed it put its records
        h->fidx = nb; h->pn = h->pn_buf[nb]; h->samples = h->samples_buf[nb];
        h->deferred = tsdf_handle::DeferredPack();
        h->pix_su = q.su; h->pix_sv = q.sv;
        h->have_frame = true;
        h->frame_serial++;
        h->frame_has_nrm = q.has_nrm;
        h->frame_has_rgb = q.has_rgb;
        return TSDF_OK;
    }
    if (q.blk >= 0) {
        // a host / depth frame whose planes sit in a block of the queue's ring: it becomes current the way a frame handed
        // over in device memory does (deferred packing), once the caller's buffers have been read
        const int blk = q.blk;
        q.blk = -1;
        if (q.direct) {
            HIP_TRY(h, hipEventSynchronize(h->ev_qblk[blk]));
        } else {
            const auto tw0 = std::chrono::steady_clock::now();
            wait_staging_job(h, q.job);                      // the staging thread is done with the caller's buffers
            if (h->sp.on) h->sp.next_wait += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - tw0).count();
            if (q.rc != TSDF_OK) { const int r = q.rc; q.rc = TSDF_OK; h->err = q.msg; return r; }
            if (q.err != hipSuccess) return fail(h, TSDF_E_HIP, "tsdf_queue_frame: staging failed: %s", hipGetErrorString(q.err));
        }
        const DevPlanes bp = block_planes(h, blk);
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_qblk[blk], 0));   // the first tracker pass and the integrate launch read the planes
        return block_frame_current(h, blk, bp, q.has_nrm, q.has_rgb, false, false);
    }
    return fail(h, TSDF_E_BADARG, "tsdf_next_frame: the queued frame has no device block (internal error)");
}

int tsdf_set_frame_device(tsdf_handle* h, const float* d_xyz, const float* d_nrm, const uint8_t* d_rgb, int32_t width, int32_t height) {
    if (!h || !d_xyz || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_set_frame_device: bad argument") : TSDF_E_BADARG;
    if (h->qcount > 0) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_set_frame_device");
    int rc = bind_device(h);
    if (rc) return rc;
    rc = ensure_frame_buffers(h, width, height, false);
    if (rc) return rc;
    h->staged_xyz = false;
    if (h->defer_device_pack) return defer_pack(h, d_xyz, d_nrm, d_rgb);
    return run_pack(h, d_xyz, d_nrm, d_rgb, false);
}

int tsdf_set_frame_aos(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L,
                       int32_t width, int32_t height) {
    if (!h || !L || (!points && !normals) || width <= 0 || height <= 0)
        return h ? fail(h, TSDF_E_BADARG, "tsdf_set_frame_aos: bad argument") : TSDF_E_BADARG;
    if (h->qcount > 0) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_set_frame_aos");
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
    const bool had_rgb = h->frame_has_rgb;
    h->staged_xyz = false;                                 // until this frame's planes are complete on the device
    // a whole new frame takes the other set of pinned planes (next_staging_set); normals alone go into the set their points
    // went through, once nothing copies out of it any more
    if (points) { rc = next_staging_set(h, npix); if (rc) return rc; }
    else HIP_TRY(h, hipStreamSynchronize(h->fstream));
    float* const px = h->pin_xyz; float* const pnm = h->pin_nrm; uint8_t* const pc = h->pin_rgb;
    const tsdf_aos_layout lay = *L;
    // a new cloud: its tracker samples go up first (the passes of a following tsdf_track run under the planes' copy)
    const bool samples_first = samples_first_enabled() && points != nullptr;
    if (points) {
        // a whole new frame: its planes go into a block of the ring and are packed by its own integrate launch (round 6)
        int blk = -1;
        DevPlanes dst;
        rc = acquire_queue_block(h, &blk, &dst);
        if (rc) return rc;
        StageFirst first;
        if (samples_first) {
            rc = ensure_pin_samples(h);
            if (rc) return rc;
            first = samples_first_work(h, points, (size_t)lay.point_stride, (size_t)lay.xyz_offset, width, h->fidx ^ 1, true);
        }
        HIP_TRY(h, stage_and_upload(h, npix, true, normals != nullptr, color, [&](size_t i0, size_t i1) {
            repack_aos(lay, points, normals, color, px, pnm, pc, i0, i1);
        }, 1, &dst, samples_first ? &first : nullptr));
        rc = staging_set_copies_issued(h);
        if (rc) return rc;
        return block_frame_current(h, blk, dst, normals != nullptr, color, samples_first, true);
    }
    if (h->staged_blk >= 0) {
        // the normals of the CURRENT frame, whose points sit in a block of the ring: into that block's normal plane; the
        // frame's integrate launch packs them (for a frame that was packed already: once more, with the normals)
        const int blk = h->staged_blk;
        const size_t plane = plane_stride_bytes(h->in_cap);
        DevPlanes dst;
        dst.xyz = reinterpret_cast<float*>(h->qblk[blk]); dst.nrm = reinterpret_cast<float*>(h->qblk[blk] + plane);
        dst.rgb = reinterpret_cast<uint8_t*>(h->qblk[blk] + 2 * plane);
        HIP_TRY(h, stage_and_upload(h, npix, false, true, false, [&](size_t i0, size_t i1) {
            repack_aos(lay, nullptr, normals, false, nullptr, pnm, nullptr, i0, i1);
        }, 1, &dst, nullptr));
        HIP_TRY(h, hipEventRecord(h->ev_frame, h->fstream));
        if (!h->deferred.pending) {
            borrow_device_frame(h, h->frame_serial, true);
            h->deferred = tsdf_handle::DeferredPack();
            h->deferred.pending = true; h->deferred.samples_listed = true;     // the sample list of this frame exists
            h->deferred.xyz = dst.xyz; h->deferred.rgb = had_rgb ? dst.rgb : nullptr;
            h->qblk_serial[blk] = h->frame_serial;
        }
        h->deferred.nrm = dst.nrm;
        h->frame_has_nrm = true;
        h->records_pending = true;                        // tsdf_integrate waits for ev_frame
        h->staged_xyz = true;
        return TSDF_OK;
    }
    return fail(h, TSDF_E_NO_FRAME, "tsdf_set_frame_aos: normals alone complete a host frame the library holds; the current frame came from device memory");
}


// ---- depth pre-processing (optional stage in front of the hot path) -------------------------------------------

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
    if (!queued && h->qcount > 0)
        return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", who);
    int rc = bind_device(h);
    if (rc) return rc;
    if (queued) { rc = queue_admit(h, true, width, height); if (rc) return rc; }
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

// Upload, back-projection, filter and normals of a depth frame on the frame stream; the planes of `out` hold the
// frame afterwards.  Runs on the caller's thread (tsdf_set_depth_frame) or on the queue's library thread
// (tsdf_queue_depth_frame): the bilateral grid's depth extent is the one host round trip of this path.
// a pageable buffer into pinned staging memory, on the staging threads (the depth + rgb of a 640x480 frame are 1.5 MB:
// ~150 us on ONE thread, most of a frame's time on the queue's library thread -- round 6)
void pooled_copy(tsdf_handle* h, void* dst, const void* src, size_t bytes) {
    HostPool* const pool = bytes >= ((size_t)1 << 17) ? host_pool(h) : nullptr;
    if (!pool) { std::memcpy(dst, src, bytes); return; }
    const std::function<void(int, int)> job = [&](int part, int parts) {
        const size_t a = (bytes * (size_t)part / (size_t)parts) & ~(size_t)63;
        const size_t b = part + 1 == parts ? bytes : (bytes * (size_t)(part + 1) / (size_t)parts) & ~(size_t)63;
        if (b > a) std::memcpy(static_cast<char*>(dst) + a, static_cast<const char*>(src) + a, b - a);
    };
    pool->run(job);
}

int depth_frame_work(tsdf_handle* h, const char* who, const uint16_t* depth16, const float* depthf, const uint8_t* rgb,
                     int32_t width, int32_t height, const tsdf_preproc_params& pp, bool* direct_out, const DevPlanes& out) {
    const size_t npix = (size_t)width * height;
    const bool use_grid = pp.grid_filter != 0 && pp.radius > 0;
    HIP_TRY(h, hipStreamSynchronize(h->fstream));          // pinned staging may still feed the previous frame
    const size_t dbytes = npix * (depth16 ? sizeof(uint16_t) : sizeof(float));
    const void* dsrc = depth16 ? (const void*)depth16 : (const void*)depthf;
    // page-locked caller buffers are copied from directly, as in tsdf_set_frame
    const bool direct = is_pinned_host(dsrc, dbytes) && (!rgb || is_pinned_host(rgb, npix * 3));
    *direct_out = direct;
    if (!direct) pooled_copy(h, h->pin_depth, dsrc, dbytes);
    HIP_TRY(h, hipMemcpyAsync(h->pre_depth, direct ? dsrc : h->pin_depth, dbytes, hipMemcpyHostToDevice, h->fstream));
    HIP_TRY(h, launch_depth_to_z(h->fstream, depth16 ? (const uint16_t*)h->pre_depth : nullptr,
                                 depth16 ? nullptr : (const float*)h->pre_depth, pp.depth_scale, (int)npix, h->pre_z,
                                 use_grid ? h->pre_minmax : nullptr));
    if (use_grid)
        HIP_TRY(h, hipMemcpyAsync(h->pin_minmax, h->pre_minmax, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, h->fstream));
    if (rgb) {
        if (!direct) pooled_copy(h, h->pin_rgb, rgb, npix * 3);
        HIP_TRY(h, hipMemcpyAsync(out.rgb, direct ? rgb : h->pin_rgb, npix * 3, hipMemcpyHostToDevice, h->fstream));
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
                              h->pre_z, h->pre_zf, out.xyz, out.nrm));
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
    // the pre-processed planes go into a block of the ring and are packed by the frame's own integrate launch (round 6)
    int blk = -1;
    DevPlanes dst;
    rc = acquire_queue_block(h, &blk, &dst);
    if (rc) return rc;
    rc = depth_frame_work(h, "tsdf_set_depth_frame", depth16, depthf, rgb, width, height, pp, &direct, dst);
    if (rc) return rc;
    rc = block_frame_current(h, blk, dst, true, rgb != nullptr, false, true);
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
    tsdf_handle::Queued& q = h->queued_slot(h->qcount);
    q = tsdf_handle::Queued();
    q.has_nrm = true; q.has_rgb = rgb != nullptr;
    // the pre-processed planes go into a block of the queue's ring and are packed by the frame's own integrate launch
    DevPlanes dst;
    rc = acquire_queue_block(h, &q.blk, &dst);
    if (rc) return rc;
    const int blk = q.blk;
    tsdf_handle::Queued* const slot = &q;
    q.job = submit_staging_job(h, [h, depth16, depthf, rgb, width, height, pp, dst, blk, slot] {
            bool direct = false;
            t_err_sink = &slot->msg;
            int r = depth_frame_work(h, "tsdf_queue_depth_frame", depth16, depthf, rgb, width, height, pp, &direct, dst);
            t_err_sink = nullptr;
            hipError_t e = hipSuccess;
            if (r == TSDF_OK) e = hipEventRecord(h->ev_qblk[blk], h->fstream);
            const bool use_grid = pp.grid_filter != 0 && pp.radius > 0;
            if (r == TSDF_OK && e == hipSuccess && direct && !use_grid) e = hipEventSynchronize(h->ev_copied);   // the caller's buffers have been read
            slot->rc = r;
            slot->err = e;
        });
    if (!q.job) return fail(h, TSDF_E_NOMEM, "tsdf_queue_depth_frame: cannot start the staging thread");
    q.active = true; h->qcount++;
    return TSDF_OK;
}

int tsdf_get_preprocessed(tsdf_handle* h, float* xyz, float* nrm) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    // (a frame queued behind the current one goes into a block of its own since round 6: the current frame's planes stay)
    if (!h->staged_xyz || !h->staged_planes[0] || !h->staged_planes[1])
        return fail(h, TSDF_E_NO_FRAME, "tsdf_get_preprocessed: the library does not hold the planes of the current frame (it came from device memory)");
    const size_t bytes = (size_t)h->fw * h->fh * 3 * sizeof(float);
    if (xyz) HIP_TRY(h, hipMemcpyAsync(xyz, h->staged_planes[0], bytes, hipMemcpyDeviceToHost, h->fstream));
    if (nrm) HIP_TRY(h, hipMemcpyAsync(nrm, h->staged_planes[1], bytes, hipMemcpyDeviceToHost, h->fstream));
    HIP_TRY(h, hipStreamSynchronize(h->fstream));
    return TSDF_OK;
}
