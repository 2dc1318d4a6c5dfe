// api_hotpath.cpp -- the per-frame hot path behind the C ABI: one Gauss-Newton pass (launch, fan-in, exchange between
// ranks), the Gauss-Newton loop (reference src/camera_tracking.cpp:79-239), the integrate launch (src/sdf.cpp:224-315),
// the reference's two calls on its own clouds, tsdf_sample (see handle.hpp).
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

void fill_integrate_params(const tsdf_handle* h, IntegrateParams& p) {
    p.g = h->grid;
    std::memcpy(p.rot_inv, h->pose.rot_inv, sizeof p.rot_inv);
    std::memcpy(p.rot_inv_trans, h->pose.rot_inv_trans, sizeof p.rot_inv_trans);
    std::memcpy(p.K, h->K, sizeof p.K);
    p.width = h->fw; p.height = h->fh;
    p.pix_su = h->pix_su; p.pix_sv = h->pix_sv;
    p.with_color = h->cfg.with_color;
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

// Launch one accumulation pass and wait for its kRedWidth-double result row in h->red_host.
// reduce_ranks: sum the leading kRedAllreduce entries over ranks (RCCL on the device buffer, or hook).
int accumulate_pass(tsdf_handle* h, bool reduce_ranks, bool later_pass) {
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
        ? reinterpret_cast<double*>(h->shm.dev_base + shm_slot_offset(h->shm, h->shm.rank, seq)) : h->red_host;
    const unsigned long long dev_word = dev_publish_shm ? shm_word(h->shm, seq) : seq;
    if (h->timing_track) HIP_TRY(h, hipEventRecord(h->ev_track.a, h->stream));
    // the row ends up on this host anyway (no in-stream all-reduce, no device-published slot): let the device stop
    // after the shard level and add the <= 8 shard rows here
    const bool host_fanin = !use_rccl && !use_peer && !dev_publish_shm && !h->timing_track;
    // device-side exchange: the workgroup that finishes the row swaps it with the other ranks before handing it out
    PeerExchange px;
    if (use_peer) px = peer_exchange_for(h, seq);
    const auto tp1 = h->track_profile ? std::chrono::steady_clock::now() : tp0;
    // passes >= 1 whose row comes to this host: through the library's own queue (pass 0 stays on the stream, ordered behind
    // the integration; a later pass is only submitted after the host has seen the row of the one before it)
    const bool via_aql = h->aql_on && h->aql.ready() && later_pass && host_fanin && !from_plane;
    bool took_queue = false;
    HIP_TRY(h, launch_track_folded(h->stream, p, h->dw, h->samples, h->partials, h->fold_ctr + (seq & 1ull) * track_fold_counter_words(), h->red_dev,
                                   use_rccl ? nullptr : host_row, host_fanin ? h->shard_host : nullptr, dev_word, seq,
                                   use_peer ? &px : nullptr, via_aql ? &h->aql : nullptr, &took_queue));
    if (took_queue) h->cnt.track_passes_own_queue++;          // counted only when the queue really took the dispatch
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
        HIP_TRY(h, launch_track_publish(h->stream, h->red_dev, h->red_host, seq));
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
            if (!h->aql.wait_idle()) {               // the queue's packet never completed: the queue is dead, later passes use the stream
                h->aql_on = false;
                return fail(h, TSDF_E_HIP, "tracker pass %llu: the library's own queue did not complete its dispatch (queue disabled)", seq);
            }
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
        std::string serr;
        const int rc2 = shm_fan_in(h->shm, seq, kRedAllreduce, h->red_host, &serr);
        if (rc2) return fail(h, rc2, "%s", serr.c_str());
        arrived = true;
    } else {
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
        shm_publish(h->shm, seq, h->red_host);
        std::string serr;
        const int rc2 = shm_fan_in(h->shm, seq, kRedAllreduce, h->red_host, &serr);
        if (rc2) return fail(h, rc2, "%s", serr.c_str());
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

}  // namespace tsdf_api

namespace {
void unpack_normal_equations(const double* row, double A[36], double b[6]) {
    int e = 0;
    for (int a = 0; a < 6; ++a)
        for (int c = a; c < 6; ++c) { A[6 * a + c] = row[e]; A[6 * c + a] = row[e]; ++e; }
    for (int a = 0; a < 6; ++a) b[a] = row[21 + a];
}
}  // namespace

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
    bool fused = false, fused_queued = false, wrote_own_list = false;
    ReleaseWord rel;                         // tells the host when the borrowed planes packed by this launch have been read
    tsdf_handle::Queued& q = h->queued_front();       // (a frame in device memory waits at the front of the queue only)
    if (h->qcount > 0 && q.active && q.device && q.deferred && !q.packed) {
        const bool own_too = h->deferred.pending;
        if (own_too) {
            PackArgs own = pack_args(h, h->deferred.xyz, h->deferred.nrm, h->deferred.rgb, h->pix_su, h->pix_sv, h->fidx);
            if (h->deferred.samples_listed) own.samples = nullptr;
            wrote_own_list = own.samples != nullptr;
            HIP_TRY(h, launch_pack(h->stream, own));
            h->deferred.pending = false;
        }
        q.su = h->pix_su; q.sv = h->pix_sv;      // laid out for this frame's pose: the next one's is close to it
        pa = pack_args(h, q.d_xyz, q.d_nrm, q.d_rgb, q.su, q.sv, h->fidx ^ 1);
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
        if (last_items != ~0ull && h->integrate_cus > 0) {
            const unsigned long long per_wg_cu = (unsigned long long)h->integrate_cus * (kIntegrateBlock / 64) * 16ull;   // items that give every wavefront 16
            const int max_per_cu = h->integrate_blocks / ((h->integrate_cus + 7) / 8 * 8) > 0 ? h->integrate_blocks / ((h->integrate_cus + 7) / 8 * 8) : 1;
            int want = (int)((last_items + per_wg_cu - 1) / per_wg_cu);
            want = want < 1 ? 1 : want > max_per_cu ? max_per_cu : want;
            blocks = (h->integrate_cus * want + 7) / 8 * 8;
            if (blocks > h->integrate_blocks) blocks = h->integrate_blocks;
        }
        rel.items_word = h->release_host + 2;
        const hipError_t le = launch_integrate(h->stream, p, h->dw, h->crgb, h->pn, h->counters, h->worklist, h->work_count,
                                               blocks, h->integrate_launches, h->wg_counts,
                                               fused ? &pa : nullptr, &rel);
        if (le != hipSuccess) {
            // nothing was packed: the frames stay borrowed and unpacked (a later launch, or tsdf_synchronize, packs them)
            if (fused) for (auto& b : h->borrowed) if (b.stream == 0 && b.ticket == rel.ticket && b.serial >= h->frame_serial + (fused_queued ? 1 : 0)) b.stream = -1;
            return fail(h, TSDF_E_HIP, "launch_integrate failed: %s (%s:%d)", hipGetErrorString(le), __FILE__, __LINE__);
        }
    }
    h->integrate_launches++;
    if (fused) {                             // sample lists this launch (or the pack launch in front of it) writes: see samples_written_ticket
        if (fused_queued) { h->samples_written_ticket[h->fidx ^ 1] = rel.ticket; if (wrote_own_list) h->samples_written_ticket[h->fidx] = rel.ticket; }
        else if (pa.samples) h->samples_written_ticket[h->fidx] = rel.ticket;
    }
    if (fused_queued) q.packed = true;       // only now: a failed launch must not leave an unpacked record buffer marked as packed
    h->deferred.pending = false;             // records and sample list of the current frame are complete from here on
    rc = timed_end(h, ep, h->stream);
    if (rc) return rc;
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

int tsdf_track(tsdf_handle* h, tsdf_track_stats* stats) {
    int rc = check_ready(h, true);
    if (rc) return rc;
    return track_loop(h, stats);
}
namespace tsdf_api {
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
}  // namespace tsdf_api

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
}  // namespace

namespace {
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
    if (h->qcount > 0) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_track_aos");
    bool color = false;
    int rc = check_point_layout(h, "tsdf_track_aos", L, &color);
    if (rc) return rc;
    if (normals) { rc = check_normal_layout(h, "tsdf_track_frame_aos", L); if (rc) return rc; }
    using pclk = std::chrono::steady_clock;
    const bool prof = h->sp.on;
    auto lap = [prof](pclk::time_point& t, double& acc) { if (prof) { const pclk::time_point n = pclk::now(); acc += std::chrono::duration<double, std::nano>(n - t).count(); t = n; } };
    pclk::time_point tp = prof ? pclk::now() : pclk::time_point();
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
    // the other set of pinned planes: the copies out of it were those of the frame before the last one
    std::swap(h->pin_xyz, h->alt_xyz); std::swap(h->pin_nrm, h->alt_nrm); std::swap(h->pin_rgb, h->alt_rgb); std::swap(h->pin_samples[0], h->pin_samples[1]);
    std::swap(h->ev_stage_done[0], h->ev_stage_done[1]); std::swap(h->stage_recorded[0], h->stage_recorded[1]);
    if (h->stage_recorded[0]) HIP_TRY(h, hipEventSynchronize(h->ev_stage_done[0]));
    lap(tp, h->sp.a_prep2);
    h->staged_xyz = false;
    h->tracked = tsdf_handle::TrackedCloud();
    // the cloud's planes go into a block of the ring of device blocks; the frame's own integrate launch packs them (round 6)
    int blk = -1;
    DevPlanes dst;
    rc = acquire_queue_block(h, &blk, &dst);
    if (rc) return rc;
    // 1. the tracker's samples, straight from the cloud, in front of everything else (upload_samples_first)
    lap(tp, h->sp.a_prep);
    rc = upload_samples_first(h, points, (size_t)L->point_stride, (size_t)L->xyz_offset, width);
    if (rc) return rc;
    lap(tp, h->sp.a_gather);
    // the frame is current from here on: sample list on its way, packing deferred, no normals yet (tsdf_integrate_aos brings them)
    rc = defer_pack(h, dst.xyz, nullptr, color ? dst.rgb : nullptr, false, true);
    if (rc) return rc;
    h->deferred.samples_listed = true;
    h->qblk_serial[blk] = h->frame_serial;
    // 2. the whole cloud -> pinned planes -> the block, on the library threads and the frame stream, under the passes
    //    (tsdf_track_frame_aos: the normals as well -- one block, one copy: the frame is complete when the passes are over and
    //    tsdf_integrate only waits for the copy on the device)
    uint64_t job = 0;
    {
        const tsdf_aos_layout lay = *L;
        h->stage_err = hipSuccess;
        job = submit_staging_job(h, [h, npix, points, normals, lay, color, dst] {
            float* const px = h->pin_xyz; float* const pnm = h->pin_nrm; uint8_t* const pc = h->pin_rgb;
            hipError_t e = stage_and_upload(h, npix, true, normals != nullptr, color, [&](size_t i0, size_t i1) {
                repack_aos(lay, points, normals, color, px, pnm, pc, i0, i1);
            }, normals ? 2 : 1, &dst);
            if (e == hipSuccess) e = hipEventRecord(h->ev_frame, h->fstream);
            h->stage_err = e;
        });
    }
    if (!job) return fail(h, TSDF_E_NOMEM, "tsdf_track_aos: cannot start the staging thread");
    lap(tp, h->sp.a_issue);
    // 3. estimate_new_position on the list
    const int rc_track = track_loop(h, stats);
    lap(tp, h->sp.a_loop);
    // 4. the cloud is the caller's again when this call returns
    wait_staging_job(h, job);
    lap(tp, h->sp.a_wait);
    if (prof) h->sp.aos_frames++;
    if (h->stage_err != hipSuccess) return fail(h, TSDF_E_HIP, "tsdf_track_aos: staging the cloud failed: %s", hipGetErrorString(h->stage_err));
    HIP_TRY(h, hipEventRecord(h->ev_stage_done[0], h->fstream));      // the copies out of this set of planes, so far
    h->stage_recorded[0] = true;
    h->staged_xyz = true; h->staged_planes[0] = dst.xyz; h->staged_planes[1] = dst.nrm; h->staged_blk = blk;
    h->tracked.valid = true; h->tracked.color = color; h->tracked.points = points; h->tracked.w = width; h->tracked.h = height;
    h->tracked.serial = h->frame_serial; h->tracked.lay = *L;
    h->tracked.normals = normals;
    if (normals && h->deferred.pending) { h->deferred.nrm = dst.nrm; h->frame_has_nrm = true; }
    h->records_pending = true;               // whoever integrates this frame waits for the planes' copy (ev_frame) on the device
    return rc_track;
}
}  // namespace

int tsdf_integrate_aos(tsdf_handle* h, const void* points, const void* normals, const tsdf_aos_layout* L, int32_t width, int32_t height,
                       tsdf_integrate_stats* stats) {
    if (!h || !L || !normals || width <= 0 || height <= 0) return h ? fail(h, TSDF_E_BADARG, "tsdf_integrate_aos: bad argument (the normals are required)") : TSDF_E_BADARG;
    if (h->qcount > 0) return fail(h, TSDF_E_BADARG, "%s: a frame is queued (tsdf_queue_frame): take it with tsdf_next_frame first", "tsdf_integrate_aos");
    bool color = false;
    int rc = points ? check_point_layout(h, "tsdf_integrate_aos", L, &color) : TSDF_OK;
    if (rc) return rc;
    rc = check_normal_layout(h, "tsdf_integrate_aos", L);
    if (rc) return rc;
    using pclk = std::chrono::steady_clock;
    const bool prof = h->sp.on;
    auto lap = [prof](pclk::time_point& t, double& acc) { if (prof) { const pclk::time_point n = pclk::now(); acc += std::chrono::duration<double, std::nano>(n - t).count(); t = n; } };
    pclk::time_point tp = prof ? pclk::now() : pclk::time_point();
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
    const bool candidate = tc.valid && h->have_frame && h->staged_xyz && h->staged_blk >= 0 && tc.serial == h->frame_serial && tc.w == width && tc.h == height &&
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
    const size_t npix = (size_t)width * height;
    const tsdf_aos_layout lay = *L;
    const int blk = h->staged_blk;
    const DevPlanes dst = block_planes(h, blk);
    HostPool* const pool = host_pool(h);
    float* const px = h->pin_xyz; float* const pnm = h->pin_nrm; uint8_t* const pc = h->pin_rgb;
    auto split = [npix](int part, int parts, size_t* i0, size_t* i1) {
        // multiples of four points, so that the 16-byte stores of the repack stay aligned in every part
        *i0 = (npix * (size_t)part / (size_t)parts) & ~(size_t)3; *i1 = part + 1 == parts ? npix : (npix * (size_t)(part + 1) / (size_t)parts) & ~(size_t)3;
    };
    // 1. the normals: repack into the pinned plane of the set that holds the cloud and copy (this copy is on the frame's
    //    critical path; repacking piece c+1 while piece c travels was measured, 8 alternations each: medians 2223 / 2186 /
    //    2224 frames/s with 1 / 2 / 3 pieces -- profiles/r05_entry_points.json -- so one piece)
    {
        constexpr int kNc = 1;
        for (int c = 0; c < kNc; ++c) {
            const size_t c0 = (npix * (size_t)c / (size_t)kNc) & ~(size_t)3, c1 = c + 1 == kNc ? npix : (npix * (size_t)(c + 1) / (size_t)kNc) & ~(size_t)3;
            const std::function<void(int, int)> fill = [&](int part, int parts) {
                const size_t n = c1 - c0;
                const size_t i0 = c0 + ((n * (size_t)part / (size_t)parts) & ~(size_t)3), i1 = part + 1 == parts ? c1 : c0 + ((n * (size_t)(part + 1) / (size_t)parts) & ~(size_t)3);
                repack_aos(lay, nullptr, normals, false, nullptr, pnm, nullptr, i0, i1);
            };
            if (pool) pool->run(fill); else fill(0, 1);
            HIP_TRY(h, hipMemcpyAsync(dst.nrm + 3 * c0, pnm + 3 * c0, (c1 - c0) * 3 * sizeof(float), hipMemcpyHostToDevice, h->fstream));
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
        }, 1, &dst));
    }
    HIP_TRY(h, hipEventRecord(h->ev_stage_done[0], h->fstream));
    h->stage_recorded[0] = true;
    // 3. SDF::update: the frame's integrate launch packs the records (cloud + the normals that have just been sent) itself, behind
    //    the copies (ev_frame); for a changed cloud it rewrites the sample list as well
    HIP_TRY(h, hipEventRecord(h->ev_frame, h->fstream));
    if (!h->deferred.pending) {                  // (packed once already -- a tsdf_synchronize in between: once more, with the normals)
        borrow_device_frame(h, h->frame_serial, true);
        h->deferred = tsdf_handle::DeferredPack();
        h->deferred.pending = true;
        h->deferred.xyz = dst.xyz; h->deferred.rgb = h->frame_has_rgb ? dst.rgb : nullptr;
        h->qblk_serial[blk] = h->frame_serial;
    }
    h->deferred.nrm = dst.nrm;
    h->deferred.samples_listed = same;
    h->frame_has_nrm = true;
    h->records_pending = true;
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
