// host_util.cpp -- see host_util.hpp.  Host code only (no HIP): builds with plain g++ and under the sanitizers.
#include "host_util.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <ctime>
#include <new>

#include "host_math.hpp"

namespace tsdf {
namespace host {

namespace {
int shm_fail(std::string* err, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (err) *err = buf;
    return code;
}
}  // namespace

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

void shm_unmap(ShmSegment* s) {
    if (!s || !s->base) return;
    munmap(s->base, s->bytes);
    // the name was unlinked by rank 0 as soon as every rank had joined (shm_rendezvous): nothing to remove
    // here, and by now the name may belong to somebody else
    s->base = s->dev_base = nullptr;
    s->nranks = 0;
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
int shm_rendezvous(const char* name, int nranks, int rank, ShmSegment* out, std::string* err) {
    if (!name || !out || nranks <= 0 || nranks > 4096 || rank < 0 || rank >= nranks) return shm_fail(err, TSDF_E_BADARG, "shm_rendezvous: bad argument");
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
        if (fd < 0) return shm_fail(err, TSDF_E_COMM, "shm_open(%s, O_EXCL) failed: %s", name, std::strerror(errno));
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return shm_fail(err, TSDF_E_COMM, "ftruncate(%s) failed", name); }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { shm_unlink(name); return shm_fail(err, TSDF_E_COMM, "mmap(%s) failed", name); }
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
                    return shm_fail(err, TSDF_E_COMM, "tsdf_comm_init_shm(%s): rank %d did not join within 20 s", name, r);
                }
                nap();
            }
        }
        shm_unlink(name);
        __atomic_store_n(shm_hdr(base, kShmHdrGo), gen, __ATOMIC_RELEASE);
    } else {
        for (;;) {
            if (expired()) return shm_fail(err, TSDF_E_COMM, "tsdf_comm_init_shm(%s): rank 0's segment did not appear within 20 s", name);
            const int fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) { nap(); continue; }
            struct stat st;
            if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { close(fd); nap(); continue; }
            void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) return shm_fail(err, TSDF_E_COMM, "mmap(%s) failed", name);
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
                            return shm_fail(err, TSDF_E_COMM, "tsdf_comm_init_shm(%s): segment was made for another number of ranks", name);
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
                    if (expired()) { munmap(base, bytes); return shm_fail(err, TSDF_E_COMM, "tsdf_comm_init_shm(%s): no go from rank 0 within 20 s", name); }
                }
                nap();
            }
            if (!restart) break;
            munmap(base, bytes);
            base = nullptr;
        }
    }
    out->nranks = nranks; out->rank = rank; out->base = base; out->dev_base = nullptr;
    out->bytes = bytes; out->header = header; out->gen = gen; out->name = name;
    return TSDF_OK;
}

void shm_publish(const ShmSegment& s, unsigned long long seq, const double* row) {
    char* slot = s.base + shm_slot_offset(s, s.rank, seq);
    std::memcpy(slot, row, kShmRowDoubles * sizeof(double));
    __atomic_store_n(reinterpret_cast<unsigned long long*>(slot + kShmRowDoubles * sizeof(double)), shm_word(s, seq), __ATOMIC_RELEASE);
}

// Shared-memory fan-in: wait for every rank's row of pass `seq`, add the leading `n` entries in rank order
// into red (the remaining entries are this rank's own).  Slots are double-buffered by pass parity: a
// rank can only overwrite its pass-s slot when publishing pass s+2, which needs everybody's pass s+1 row,
// which nobody publishes before having read all pass-s rows.
int shm_fan_in(const ShmSegment& s, unsigned long long seq, int n, double* red, std::string* err) {
    double sum[kShmRowDoubles];
    for (int e = 0; e < kShmRowDoubles; ++e) sum[e] = 0.0;
    if (n < 0 || n > kShmRowDoubles) return shm_fail(err, TSDF_E_BADARG, "shm_fan_in: n = %d", n);
    const unsigned long long want = shm_word(s, seq);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < s.nranks; ++r) {
        const char* slot = s.base + shm_slot_offset(s, r, seq);
        const volatile unsigned long long* word =
            reinterpret_cast<const volatile unsigned long long*>(slot + kShmRowDoubles * sizeof(double));
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) break;
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
                return shm_fail(err, TSDF_E_COMM, "shared-memory fan-in: rank %d did not publish pass %llu within 20 s", r, seq);
        }
        const double* row = reinterpret_cast<const double*>(slot);
        for (int e = 0; e < n; ++e) sum[e] += row[e];
        if (r == s.rank) std::memcpy(red, row, kShmRowDoubles * sizeof(double));
    }
    std::memcpy(red, sum, (size_t)n * sizeof(double));
    return TSDF_OK;
}

}  // namespace host
}  // namespace tsdf

// ---- the part of the C ABI that needs no device ---------------------------------------------------------------------
using namespace tsdf;

extern "C" {

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
    c->slab_x0 = 0; c->slab_x1 = 0; c->halo = 0; c->device = 0; c->slab_stride = 0;
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

int tsdf_slab_range(int32_t m, int32_t nranks, int32_t rank, int32_t* x0, int32_t* x1) {
    if (m <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || !x0 || !x1) return TSDF_E_BADARG;
    const int64_t q = m / nranks, r = m % nranks;      // the first r ranks get one extra layer
    const int64_t lo = q * rank + (rank < r ? rank : r);
    *x0 = (int32_t)lo;
    *x1 = (int32_t)(lo + q + (rank < r ? 1 : 0));
    return TSDF_OK;
}

// Block-cyclic placement (tsdf_config::slab_stride): rank r of N owns the blocks [r B + j N B, (r+1) B + j N B).  block = 0:
// B = m / (2 N) rounded down to a power of two -- two blocks per rank, where the halo's work ((B + 2 halo) / B) and the balance
// meet -- doubled until N B >= B + 2 halo.  TSDF_E_BADARG when no block fits (m not a power of two, the halo
// too wide for the number of ranks, more ranks than blocks).
int tsdf_cyclic_range(int32_t m, int32_t nranks, int32_t rank, int32_t halo, int32_t block, int32_t* x0, int32_t* x1, int32_t* stride) {
    if (m <= 0 || (m & (m - 1)) != 0 || nranks < 2 || rank < 0 || rank >= nranks || halo < 0 || block < 0 || !x0 || !x1 || !stride) return TSDF_E_BADARG;
    int64_t B = block;
    if (B == 0) {
        B = 1;
        while (2 * B <= m / (2 * (int64_t)nranks)) B *= 2;
        while (B < m && (int64_t)(nranks - 1) * B < 2 * (int64_t)halo) B *= 2;
    }
    if ((B & (B - 1)) != 0 || m % B != 0 || (int64_t)nranks * B > m || (int64_t)(nranks - 1) * B < 2 * (int64_t)halo) return TSDF_E_BADARG;
    *x0 = (int32_t)(rank * B); *x1 = (int32_t)((rank + 1) * B); *stride = (int32_t)(nranks * B);
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
    // a degenerate volume or camera would put NaN / inf into the weights (rejected far from the cause by
    // tsdf_slab_range_weighted) or silently clip the frustum to nothing (fx or fy == 0): refuse here
    if (!(c->width > 0.f) || !(c->height > 0.f) || !(c->depth > 0.f) || !std::isfinite(c->width) || !std::isfinite(c->height) || !std::isfinite(c->depth) ||
        !std::isfinite(max_depth)) return TSDF_E_BADARG;
    for (int a = 0; a < 9; ++a) if (!std::isfinite(K[a]) || !std::isfinite(rot[a])) return TSDF_E_BADARG;
    for (int a = 0; a < 3; ++a) if (!std::isfinite(trans[a]) || !std::isfinite(c->origin[a])) return TSDF_E_BADARG;
    if (K[0] == 0.0 || K[4] == 0.0) return TSDF_E_BADARG;
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

}  // extern "C"
