// track_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the tracker.
//
//   track_kernel       one Gauss-Newton accumulation pass    (reference src/camera_tracking.cpp:146-189,
//                      + get_partial_derivative :246-363, SDF::interpolate_distance sdf.cpp:127-163)
//                      incl. the fixed-order fan-in of the per-workgroup partial normal equations (one launch per pass)
//   sample_kernel      SDF::interpolate_distance batched (tsdf_sample)
//   peer_exchange_kernel / track_publish_kernel   hand-off of the reduced row between ranks / to the host
//
// Built twice from this one file with the same flags: into libtsdf_hip.so, and as the stand-alone code object
// lib/tsdf_track.hsaco from which the library's own AQL queue (csrc/aql_queue.cpp) dispatches track_kernel.
// MUST be compiled with -ffp-contract=off and without fast-math (see integrate_kernels.hip).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "aql_queue.hpp"
#include "device_util.h"
#include "tsdf_device.h"

#ifndef TSDF_BUILD_ID
#define TSDF_BUILD_ID "unversioned"      // the Makefile passes a hash of this file, its headers and the flags
#endif
// In the stand-alone code object (lib/tsdf_track.hsaco): what AqlQueue::init compares with the library's own
// track_kernel_build_id() before it dispatches anything from it.
extern "C" __device__ __attribute__((used)) const char tsdf_track_build_id[40] = TSDF_BUILD_ID;

namespace tsdf {

// ------------------------------------------------------------------------------------------------
// SDF::interpolate_distance (sdf.cpp:127-163) on the device layout.
// Returns false when no corner is valid (reference: is_interpolated = false, value NaN).
// `viol` is raised when a corner lies inside the grid but outside this rank's stored layers.

struct Vol {
    const float2* dw;
    int m, xs, xe;           // the stored x layers a look-up may touch: the slab's -- or, block-cyclic, those of ONE block (make_vol)
    int xoff;                // global layer of local layer 0 as seen from that block (the slab's xs)
    int n_layers;            // all stored layers of the handle
    __device__ __forceinline__ long long dummy() const { return (long long)n_layers * m * m; }   // the {0,0} pair behind the volume
};
// The part of the volume a look-up around the centre voxel layer `ci` may touch.  A plain slab: all of it.  Block-cyclic: the
// stored layers of the block `ci` belongs to (a sample this rank owns has its 13 look-ups within the halo of that block).
__device__ __forceinline__ Vol make_vol(const Grid& g, const float2* dw, int ci) {
    if (g.blk_own <= 0) return Vol{dw, g.m, g.xs, g.xe, g.xs, g.xe - g.xs};
    const int b = grid_block_of(g, ci);
    const int first = g.blk_first + b * g.blk_stride;
    const int lo = first < 0 ? 0 : first, hi = first + g.blk_layers > g.m ? g.m : first + g.blk_layers;
    return Vol{dw, g.m, lo, hi, first - b * g.blk_layers, g.n_blocks * g.blk_layers};
}

typedef float vol_f4 __attribute__((ext_vector_type(4), aligned(8)));   // two neighbouring voxels {D,W,D,W}

// One look-up in two halves, both straight-line code: lookup_issue computes the addresses and requests the data,
// lookup_finish runs the reference's accumulation.  With branches around the loads (as the first version had) hipcc
// put an s_waitcnt vmcnt(0) after every one of the four row loads -- four serialized round trips per look-up,
// eight for a lane with two look-ups; branch-free, all loads of a lane are in flight together.
// The corners k and k+1 of one (i,j) voxel row are neighbours in memory: ONE 16-byte load per row instead of two
// 8-byte ones.
struct Lookup {
    float fi, fj, fk;
    int bi, bj, bk;
    bool k_ok[2];            // corner k / k+1 inside the grid in k
    vol_f4 v[4];             // {D,W} of corner k (x,y) and k+1 (z,w); rows that are not stored hold the dummy pair: W = 0
};

// The pair is read at k = bk clamped to [-1, m-1]: at k = -1 / m-1 one half is the last / first voxel of the
// neighbouring row (or the padding around the volume) and k_ok masks it.
__device__ __forceinline__ void lookup_issue(const Vol& V, double vx, double vy, double vz, Lookup& L, unsigned& viol) {
    L.fi = (float)vx; L.fj = (float)vy; L.fk = (float)vz;                // f64 -> f32, sdf.cpp:130-132
    L.bi = trunc_x86(L.fi); L.bj = trunc_x86(L.fj); L.bk = trunc_x86(L.fk);
    const int bk = L.bk;
    int kc = bk < -1 ? -1 : bk;
    kc = kc > V.m - 1 ? V.m - 1 : kc;
    // INT_MIN + 1 wraps nowhere: bk + 1 is only compared
    L.k_ok[0] = (bk >= 0) & (bk < V.m);
    L.k_ok[1] = (bk >= -1) & (bk < V.m - 1);
    // the four voxel rows (i,j), (i,j+1), (i+1,j), (i+1,j+1): validity per axis, one 64-bit base address
    const int m = V.m, bi = L.bi, bj = L.bj;
    const bool i_in[2] = {bi >= 0 && bi < m, bi >= -1 && bi < m - 1};                       // sdf.h:113-119
    const bool j_in[2] = {bj >= 0 && bj < m, bj >= -1 && bj < m - 1};
    const bool i_st[2] = {i_in[0] && bi >= V.xs && bi < V.xe, i_in[1] && bi + 1 >= V.xs && bi + 1 < V.xe};
    const bool k_any = L.k_ok[0] | L.k_ok[1];
    const long long mm = (long long)m * m;
    const long long base = (((long long)bi - V.xoff) * m + bj) * m + kc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int io = r >> 1, jo = r & 1;
        const bool in_grid = i_in[io] & j_in[jo];
        const bool stored = i_st[io] & j_in[jo];
        // the row is in the grid but not stored here: a violation if one of its two corners is in the grid
        viol |= (in_grid & !stored & k_any) ? 1u : 0u;
        const long long at = stored ? base + (io ? mm : 0ll) + (jo ? (long long)m : 0ll) : V.dummy();
        L.v[r] = *reinterpret_cast<const vol_f4*>(reinterpret_cast<const float*>(V.dw + at));
    }
}

// 1.0f / v, correctly rounded, for v in (1e-5, 4]: the fused-multiply-add core of the compiler's own f32 division
// (rcp, two refinements of the reciprocal, quotient, two residual corrections) without its range scaling and
// special-case fix-up, which do nothing in this range -- same bits, 7 instructions instead of 12.  For any other v
// (0, NaN) the value is garbage and the caller discards it.
__device__ __forceinline__ float recip_ieee_small(float v) {
    float r = __builtin_amdgcn_rcpf(v);
    const float e0 = __builtin_fmaf(-v, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = r;                                            // 1.0f * r
    const float e1 = __builtin_fmaf(-v, q, 1.0f);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-v, q, 1.0f);
    return __builtin_fmaf(e2, r, q);
}

// The reference's loop (sdf.cpp:139-162) without a branch: every corner is evaluated, skipped ones add +0.0f (the
// sums start at +0.0f and can never become -0.0f, so that changes no bit), the exact-hit early return becomes a
// latched flag.  (double)volume < 0.00001 is volume <= 1e-5f: 1e-5f is the largest float below the double constant.
__device__ __forceinline__ bool lookup_finish(const Lookup& L, float& out) {
    const float di[2] = {fabsf((float)L.bi - L.fi), fabsf((float)(L.bi + 1) - L.fi)};
    const float dj[2] = {fabsf((float)L.bj - L.fj), fabsf((float)(L.bj + 1) - L.fj)};
    const float dk[2] = {fabsf((float)L.bk - L.fk), fabsf((float)(L.bk + 1) - L.fk)};
    float w_sum = 0.0f, sum_d = 0.0f, hit_val = 0.0f;
    bool any = false, hit = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = q >> 1, u = q & 1;
        const float volume = (di[q >> 2] + dj[(q >> 1) & 1]) + dk[u];
        const float cd = u == 0 ? L.v[r].x : L.v[r].z;
        const float cw = u == 0 ? L.v[r].y : L.v[r].w;
        const bool take = L.k_ok[u] & (cw > 0.0f) & !hit;
        const bool exact = take & (volume <= 1.0e-5f);
        const bool acc = take & !exact;
        const float w = recip_ieee_small(volume);
        w_sum += acc ? w : 0.0f;
        sum_d += acc ? w * cd : 0.0f;
        hit_val = exact ? cd : hit_val;
        any |= take;
        hit |= exact;
    }
    out = hit ? hit_val : sum_d / w_sum;
    return any;
}

__device__ __forceinline__ bool interp(const Vol& V, double vx, double vy, double vz, float& out, unsigned& viol) {
    Lookup L;
    lookup_issue(V, vx, vy, vz, L, viol);
    return lookup_finish(L, out);
}

__global__ __launch_bounds__(256) void sample_kernel(Grid g, const float2* __restrict__ dw,
                                                      const double* __restrict__ vox, int n,
                                                      float* __restrict__ val, int* __restrict__ okv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double vx0 = vox[3 * t + 0];
    const Vol V = make_vol(g, dw, (vx0 >= 0.0 && vx0 < (double)g.m) ? (int)vx0 : 0);
    float out = 0.0f;
    unsigned viol = 0;
    const bool ok = interp(V, vx0, vox[3 * t + 1], vox[3 * t + 2], out, viol);
    val[t] = out;
    okv[t] = viol ? -1 : (ok ? 1 : 0);
}

hipError_t launch_sample(hipStream_t s, const Grid& g, const float2* dw, const double* vox, int32_t n,
                         float* val, int32_t* ok) {
    if (n <= 0) return hipSuccess;
    sample_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(g, dw, vox, n, val, ok);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tracker: one Gauss-Newton accumulation pass (reference src/camera_tracking.cpp:146-189 +
// get_partial_derivative :246-363).
//
// The reference does 13 dependent look-ups per sampled pixel, one after the other.  One thread per
// sample (round-1 v1) therefore means 13 serial memory round trips per wavefront and only ~535
// wavefronts for the whole chip: pure latency, 48 us per pass.  v2 spends 16 lanes per sample:
//
//   lane q of a 16-lane group   q = 0       centre voxel            -> r            (:269)
//                               q = 1..6    centre +- v_h e_k       -> J[0..2]      (:273-316)
//                               q = 7..12   (I +- w_h [e_k]x) rot p -> J[3..5]      (:318-361)
//   so the 13 look-ups of a sample are ONE memory round trip, the chip holds ~8 wavefronts per SIMD,
//   and the reference's early exits become an AND over the group (a failed look-up drops the sample
//   either way, so evaluating the others changes nothing).
//
// Stale carry-over (:156-159,176-182,261-268): an out-of-grid pixel re-adds the previous successful
// pixel's terms, i.e. a successful sample counts 1 + #{out-of-grid samples between it and the next
// in-grid one, NaN samples skipped} times.  Classification needs geometry only, so every workgroup
// classifies a window of 64 samples starting at its own kSamplesPerBlock samples (64-bit ballots = 64 consecutive
// samples of the reference's column-major visiting order) and reads the run lengths off the masks,
// looking further ahead cooperatively in the rare case a run outlives the window.
//
// Work split: 8 lanes per sample, lane q < 7 does look-up q and (q < 6) look-up 7 + q -- two look-ups = eight
// 16-byte gathers in flight per lane; 4280 wavefronts for 640x480, all resident at once (16 lanes with one
// look-up each needed 8560 wavefronts: a second, nearly empty round on 256 CUs x 32 waves).
// Reduction: lane q < 6 of a group forms J[q] J[(q+d)%6] (d = 0..3: all 21 unique products) and
// r J[q]; the 8 groups of a wavefront are added by shuffles, the 4 wavefronts through LDS, one row of
// `partials` per workgroup; the in-launch fan-in (TrackFold, below) adds the rows in a fixed order (bitwise
// reproducible).

enum { kClsSkip = 0, kClsOog = 1, kClsIn = 2 };
constexpr int kLanesPerSample = 8;
constexpr int kSamplesPerBlock = kTrackBlock / kLanesPerSample;   // 48

struct SampleGeom {
    double px, py, pz;   // camera-frame point
    double vx, vy, vz;   // continuous voxel coordinates of its world position
};

__device__ __forceinline__ int classify_sample(const TrackParams& p, const float4 s, bool exists, SampleGeom& sg) {
    sg.px = sg.py = sg.pz = 0.0; sg.vx = sg.vy = sg.vz = 0.0;
    if (!exists) return kClsSkip;
    if (is_nan(s.x) || is_nan(s.y) || is_nan(s.z)) return kClsSkip;          // camera_tracking.cpp:168
    sg.px = (double)s.x; sg.py = (double)s.y; sg.pz = (double)s.z;
    // project_camera_to_world (:55-58) + get_voxel_coordinates (sdf.h:143-147)
    const double wx = row3(&p.rot[0], sg.px, sg.py, sg.pz) + p.trans[0];
    const double wy = row3(&p.rot[3], sg.px, sg.py, sg.pz) + p.trans[1];
    const double wz = row3(&p.rot[6], sg.px, sg.py, sg.pz) + p.trans[2];
    sg.vx = (wx - p.g.origin[0]) * (double)p.g.m_div_w - 0.5;
    sg.vy = (wy - p.g.origin[1]) * (double)p.g.m_div_h - 0.5;
    sg.vz = (wz - p.g.origin[2]) * (double)p.g.m_div_d - 0.5;
    const double dm = (double)p.g.m;
    if (sg.vx < 0 || sg.vy < 0 || sg.vz < 0) return kClsOog;                 // :261-264
    if (sg.vx >= dm || sg.vy >= dm || sg.vz >= dm) return kClsOog;           // :265-268
    return kClsIn;
}

// Sample n of the reference's visiting order (columns outer, rows inner, both with the pixel stride): from the list
// pack_kernel wrote, or -- frames handed over in device memory, whose packing is deferred to the integrate launch --
// straight from the caller's xyz plane (three 4-byte loads; workgroup-uniform choice).
__device__ __forceinline__ float4 load_sample(const TrackParams& p, const float4* __restrict__ samples, int n) {
    if (p.xyz_plane) {
        const int ci = n / p.nrows, rj = n - ci * p.nrows;
        const float* __restrict__ s = p.xyz_plane + 3 * ((size_t)(rj * p.pixel_stride) * (size_t)p.plane_width + (size_t)(ci * p.pixel_stride));
        return make_float4(s[0], s[1], s[2], 0.0f);
    }
    return samples[n];
}

__device__ __forceinline__ int classify(const TrackParams& p, const float4* __restrict__ samples, int n, SampleGeom& sg) {
    const bool exists = n < p.n_samples;
    const float4 s = exists ? load_sample(p, samples, n) : make_float4(0.f, 0.f, 0.f, 0.f);
    return classify_sample(p, s, exists, sg);
}

__device__ __forceinline__ void voxel_of(const TrackParams& p, const double* R, const SampleGeom& sg,
                                         double& vx, double& vy, double& vz) {
    const double wx = row3(&R[0], sg.px, sg.py, sg.pz) + p.trans[0];
    const double wy = row3(&R[3], sg.px, sg.py, sg.pz) + p.trans[1];
    const double wz = row3(&R[6], sg.px, sg.py, sg.pz) + p.trans[2];
    vx = (wx - p.g.origin[0]) * (double)p.g.m_div_w - 0.5;
    vy = (wy - p.g.origin[1]) * (double)p.g.m_div_h - 0.5;
    vz = (wz - p.g.origin[2]) * (double)p.g.m_div_d - 0.5;
}

// End (exclusive, as a sample index) of the OpenMP column chunk that holds sample n.  The reference's carry state
// (is_interpolated, SDF_derivative, int_dist) is thread-local and starts fresh in every thread (camera_tracking.cpp:
// 148-159); `#pragma omp for` over the ncols image columns (:160-162) with GCC's default static schedule gives thread
// t < r = ncols % np the columns [t (q+1), (t+1)(q+1)) and the others q = ncols / np columns each.  A run of
// out-of-grid samples therefore never extends past the end of its chunk.  Geometry only: the same on every rank.
__device__ __forceinline__ int chunk_end_sample(const TrackParams& p, int n) {
    if (p.carry_threads <= 1) return p.n_samples;
    const int col = n / p.nrows;
    const int q = p.ncols / p.carry_threads, r = p.ncols % p.carry_threads;
    const int big = r * (q + 1);
    const int end_col = col < big ? (col / (q + 1) + 1) * (q + 1) : big + ((col - big) / q + 1) * q;   // (col >= big implies q >= 1)
    return (end_col < p.ncols ? end_col : p.ncols) * p.nrows;
}

struct TrackFold {               // in-launch fan-in of the per-workgroup rows
    unsigned* ctr;               // kTrackShards shard counters + 1 top counter, one 128-byte line each, zero between passes
    double* shard_rows;          // kTrackShards x kPartWidth
    double* red_dev;             // kRedWidth: result row for an in-stream all-reduce (may be null)
    double* host_row;            // pinned host (or shared-segment alias): kRedWidth doubles + the word (may be null)
    double* host_shards;         // pinned host, kTrackShards x kShardSlotDoubles: when given, the shard rows go to the host
                                 // (each behind its own word) and the second fan-in level runs there
    unsigned long long word;     // what is released behind host_row once it is complete
    double tag;                  // pass number carried by every row (last column): a stale row cannot pass for a fresh one
    PeerExchange peers;          // n > 0: the finished row is exchanged with the other ranks before it is handed out
};

// sc1 (device-scope, L1-bypassing, write-through) accesses for data handed from one workgroup to another inside a
// launch: the per-CU vector L1 is never refreshed by other CUs' stores and the per-XCD L2s are not coherent
// (MI355X_MICROARCH.md, inter-workgroup visibility).
__device__ __forceinline__ void store_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wavefront exchanges this rank's row (res, kRedWidth doubles in LDS) with the other ranks of the node and leaves
// the sum of the leading n_sum entries over ranks, in rank order, in res.  False: a rank did not show up in time.
// System-scope (sc0 sc1) stores and loads on uncached memory: neither this device's L2s nor a peer's hold a copy.
__device__ __forceinline__ bool peer_exchange_row(const PeerExchange& px, double* res, int n_sum, int lane) {
    const size_t mine = ((size_t)px.rank * 2 + px.parity) * kPeerSlotBytes;
    for (int r = 0; r < px.n; ++r) {
        double* slot = reinterpret_cast<double*>(px.bases[r] + mine);
        if (lane < kRedWidth) __hip_atomic_store(&slot[lane], res[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int r = lane; r < px.n; r += 64)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(px.bases[r] + mine + kRedWidth * sizeof(double)), px.word,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // everybody's row of this pass in this rank's own buffer
    const char* own = px.bases[px.rank];
    const long long t0 = wall_clock64();
    bool ok = true;
    for (int r = lane; r < px.n; r += 64) {
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(
            own + ((size_t)r * 2 + px.parity) * kPeerSlotBytes + kRedWidth * sizeof(double));
        unsigned spins = 0;                     // second bound, should the clock not be what it is expected to be
        while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != px.word) {
            if (wall_clock64() - t0 > px.timeout_ticks || ++spins > (1u << 25)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    ok = __all(ok ? 1 : 0) != 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (ok && lane < n_sum) {
        double v = 0.0;
        for (int r = 0; r < px.n; ++r)
            v += __hip_atomic_load(reinterpret_cast<const double*>(own + ((size_t)r * 2 + px.parity) * kPeerSlotBytes) + lane,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        res[lane] = v;
    }
    return ok;
}

__global__ __launch_bounds__(64) void peer_exchange_kernel(PeerExchange px, double* __restrict__ red_dev, int n_sum,
                                                           double* __restrict__ host_row, unsigned long long host_word) {
    __shared__ double s_row[kRedWidth];
    const int lane = threadIdx.x;
    if (lane < kRedWidth) s_row[lane] = red_dev[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const bool ok = peer_exchange_row(px, s_row, n_sum, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < kRedWidth) {
        double v = s_row[lane];
        if (!ok && lane == 27) v = __longlong_as_double((long long)kRowPoisonPeerTimeout);
        red_dev[lane] = v;
        if (host_row) host_row[lane] = v;
    }
    if (host_row) {
        __threadfence_system();
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_row + kRedWidth), host_word, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_peer_exchange(hipStream_t s, const PeerExchange& px, double* red_dev, int n_sum, double* host_row,
                                unsigned long long host_word) {
    if (px.n <= 0 || px.n > kPeerMaxRanks || n_sum < 0 || n_sum > kRedWidth) return hipErrorInvalidValue;
    peer_exchange_kernel<<<dim3(1), dim3(64), 0, s>>>(px, red_dev, n_sum, host_row, host_word);
    return hipGetLastError();
}

template <bool CYC /* block-cyclic placement */>
__device__ __forceinline__ void track_body(const TrackParams& p, const float2* __restrict__ dw,
                                           const float4* __restrict__ samples,
                                           double* __restrict__ partials, const TrackFold& fold) {
    constexpr int NW = kTrackBlock / 64;
    __shared__ unsigned long long s_in[1], s_oog[1];       // the 64-sample window of this workgroup
    __shared__ unsigned long long s_in2[NW], s_oog2[NW];   // look-ahead windows
    __shared__ double s_red[NW][8][8];                     // [wave][q][slot]
    __shared__ double s_rpm[54];                           // the six perturbed rotations, for lane-indexed access

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = blockIdx.x * kSamplesPerBlock;        // first of this workgroup's own samples
    // p.rpm[9*q] with a per-lane q is a vector load from the kernel-argument segment: a memory round trip in
    // front of the look-ups.  Stage the matrices in LDS while the samples are being classified.
    if (tid < 54) s_rpm[tid] = p.rpm[tid];

    // ---- phase A: the first wavefront classifies the 64-sample window [base, base+64): the workgroup's own samples
    // and the ones right after them (where the run of an own sample usually ends); the kernel is bound by instruction
    // issue, so the other wavefronts do not repeat this for samples that are rarely needed
    __shared__ double s_geom[kSamplesPerBlock][6];         // geometry + class of the own samples, handed over by the
    __shared__ int s_cls[kSamplesPerBlock];                // threads that classify them
    if (wv == 0) {
        SampleGeom win;
        const bool exists = base + lane < p.n_samples;
        const float4 smp = exists ? load_sample(p, samples, base + lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int wcls = classify_sample(p, smp, exists, win);
        // first pass over a frame whose packing is deferred: leave the own samples in the list for the passes after it
        // (a plane read is one cold line per lane, 23 KB apart; the list is 16 contiguous bytes per sample)
        // (written THROUGH, device scope: the next pass may come off another queue -- AqlQueue -- before this kernel's
        // end-of-kernel release has happened; this wavefront drains its stores before the workgroup arrives)
        if (p.xyz_plane && p.sample_list_out && exists && lane < kSamplesPerBlock) {
            unsigned long long* out = reinterpret_cast<unsigned long long*>(&p.sample_list_out[base + lane]);
            unsigned long long lo, hi;
            __builtin_memcpy(&lo, &smp.x, 8); __builtin_memcpy(&hi, &smp.z, 8);
            __hip_atomic_store(out, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(out + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < kSamplesPerBlock) {
            s_geom[lane][0] = win.px; s_geom[lane][1] = win.py; s_geom[lane][2] = win.pz;
            s_geom[lane][3] = win.vx; s_geom[lane][4] = win.vy; s_geom[lane][5] = win.vz;
            s_cls[lane] = wcls;
        }
        const unsigned long long b_in = __ballot(wcls == kClsIn);
        const unsigned long long b_oog = __ballot(wcls == kClsOog);
        if (lane == 0) { s_in[0] = b_in; s_oog[0] = b_oog; }
    }
    __syncthreads();

    // ---- phase B: this thread's own sample (group g; no second trip to memory, no second classification) and
    // its look-ups (q and q + 7)
    const int g = tid >> 3, q = tid & 7;
    const int n = base + g;
    SampleGeom sg;
    sg.px = s_geom[g][0]; sg.py = s_geom[g][1]; sg.pz = s_geom[g][2];
    sg.vx = s_geom[g][3]; sg.vy = s_geom[g][4]; sg.vz = s_geom[g][5];
    const int cls = s_cls[g];

    // stale-carry multiplicity of sample g: out-of-grid samples between it and the next in-grid one
    unsigned mult = 1;
    if (p.stale_carry) {
        // (workgroup-uniform) does the run of the last own in-grid sample reach past the window?
        constexpr unsigned long long kOwnMask = kSamplesPerBlock >= 64 ? ~0ull : ((1ull << (kSamplesPerBlock & 63)) - 1ull);
        static_assert(kSamplesPerBlock <= 64, "the own samples must fit the first ballot word");
        const unsigned long long own_in = s_in[0] & kOwnMask;
        bool need_tail = false;
        unsigned tail = 0;
        int tail_limit = p.n_samples;
        if (own_in) {
            // it reaches the window end iff no in-grid bit follows it and its column chunk goes on behind the window
            const int last_own = 63 - __clzll((long long)own_in);
            const unsigned long long above = last_own == 63 ? 0ull : ~0ull << (last_own + 1);
            tail_limit = chunk_end_sample(p, base + last_own);
            need_tail = (s_in[0] & above) == 0ull && tail_limit > base + 64;
        }
        if (need_tail) {
            bool found = false;
            for (int pos = base + 64; !found && pos < tail_limit; pos += kTrackBlock) {
                SampleGeom tmp;
                const int c2 = classify(p, samples, pos + tid, tmp);
                const bool inside = pos + tid < tail_limit;         // the run ends with its chunk
                const unsigned long long i2 = __ballot(inside && c2 == kClsIn);
                const unsigned long long o2 = __ballot(inside && c2 == kClsOog);
                __syncthreads();                        // previous round's readers are done
                if (lane == 0) { s_in2[wv] = i2; s_oog2[wv] = o2; }
                __syncthreads();
                for (int w = 0; w < NW && !found; ++w) {
                    const unsigned long long mi = s_in2[w], mo = s_oog2[w];
                    if (mi) {
                        const int nxt = __ffsll((long long)mi) - 1;
                        tail += __popcll(mo & ((1ull << nxt) - 1ull));
                        found = true;
                    } else {
                        tail += __popcll(mo);
                    }
                }
            }
        }
        if (cls == kClsIn) {
            // window positions of the sample's own column chunk
            const int lim = chunk_end_sample(p, n) - base;
            const unsigned long long chunk = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
            const unsigned long long above = (g == 63 ? 0ull : ~0ull << (g + 1)) & chunk;
            unsigned cnt = 0;
            const unsigned long long mi = s_in[0] & above;
            if (mi) {
                const int nxt = __ffsll((long long)mi) - 1;
                cnt = __popcll(s_oog[0] & above & ((1ull << nxt) - 1ull));
            } else {
                cnt = __popcll(s_oog[0] & above) + (lim > 64 ? tail : 0u);
            }
            mult = 1u + cnt;
        }
    }

    // ---- the look-ups of this lane (camera_tracking.cpp:269-361): slot A = look-up q (centre, +x -x +y -y +z -z)
    // on lanes 0..6, slot B = look-up 7 + q (r1p r1m r2p r2m r3p r3m) on lanes 0..5
    bool owned;
    int centre_layer = 0;
    if (!CYC) owned = (cls == kClsIn) && (sg.vx >= (double)p.g.own_x0) && (sg.vx < (double)p.g.own_x1);
    else { centre_layer = cls == kClsIn ? (int)sg.vx : 0; owned = (cls == kClsIn) && grid_owns_layer(p.g, centre_layer); }   // (0 <= vx < m: truncation = floor)
    float valA = 0.0f, valB = 0.0f;
    unsigned viol = 0;
    bool okA = false, okB = false;
    {
        // both look-ups are issued before either is evaluated; lanes without a look-up run the same code on their
        // sample's centre (in-cache, results masked) so that the whole section stays one basic block
        const bool actA = owned && q < 7, actB = owned && q < 6;
        const Vol V = CYC ? make_vol(p.g, dw, centre_layer) : Vol{dw, p.g.m, p.g.xs, p.g.xe, p.g.xs, p.g.xe - p.g.xs};
        double ax = sg.vx, ay = sg.vy, az = sg.vz;
        // (lanes without a look-up repeating the address of a NEIGHBOURING lane of their quad instead of the centre's:
        // measured in round 6, no difference -- profiles/r06_track_duplicate_addresses.json)
        const int qA = q < 7 ? q : 0, qB = q < 6 ? q : 0;
        if (qA >= 1) {
            const int a = (qA - 1) >> 1;
            const double step = ((qA - 1) & 1) ? -(double)p.v_h : (double)p.v_h;
            ax += (a == 0) ? step : 0.0; ay += (a == 1) ? step : 0.0; az += (a == 2) ? step : 0.0;
        }
        double bx, by, bz;
        voxel_of(p, &s_rpm[9 * qB], sg, bx, by, bz);
        Lookup LA, LB;
        unsigned violA = 0u, violB = 0u;
        lookup_issue(V, ax, ay, az, LA, violA);
        lookup_issue(V, bx, by, bz, LB, violB);
        const bool fa = lookup_finish(LA, valA), fb = lookup_finish(LB, valB);
        okA = actA && fa; okB = actB && fb;
        viol = (actA ? violA : 0u) | (actB ? violB : 0u);
        if (!actA) valA = 0.0f;
        if (!actB) valB = 0.0f;
    }
    const int gl = lane & 56;                                   // first lane of this group in the wave
    const unsigned long long maskA = __ballot(okA), maskB = __ballot(okB);
    const bool all_ok = (((maskA >> gl) & 0x7Full) == 0x7Full) && (((maskB >> gl) & 0x3Full) == 0x3Full);   // the 13 look-ups
    const unsigned long long violmask = __ballot(viol != 0u);
    const bool any_viol = ((violmask >> gl) & 0xFFull) != 0ull;

    // ---- J[q] on lanes 0..5 of the group, from the +/- partners (float quotient widened, :286,331)
    const float r0 = __shfl(valA, gl);
    const int qa = q < 6 ? q : 0;
    const int pt = qa < 3 ? 1 + 2 * qa : 2 * (qa - 3);          // lane of the + partner (slot A for q < 3, slot B after)
    const float fpA = __shfl(valA, gl + pt), fmA = __shfl(valA, gl + pt + 1);
    const float fpB = __shfl(valB, gl + pt), fmB = __shfl(valB, gl + pt + 1);
    const float fp = qa < 3 ? fpA : fpB, fm = qa < 3 ? fmA : fmB;
    const float h = qa == 0 ? p.vh2[0] : (qa == 1 ? p.vh2[1] : (qa == 2 ? p.vh2[2] : p.wh2));
    const double Jq = (double)((fp - fm) / h);
    const double J1 = __shfl(Jq, gl + (qa + 1) % 6);
    const double J2 = __shfl(Jq, gl + (qa + 2) % 6);
    const double J3 = __shfl(Jq, gl + (qa + 3) % 6);

    double acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0;
    const bool contributes = all_ok && !any_viol;
    if (contributes && q < 6) {
        const double mu = (double)mult;
        acc[0] = mu * (Jq * Jq);                                // :181  J J^T, products (q, (q+d)%6)
        acc[1] = mu * (Jq * J1);
        acc[2] = mu * (Jq * J2);
        acc[3] = mu * (Jq * J3);                                // (q >= 3 duplicates q-3; dropped by the final kernel)
        acc[4] = mu * ((double)r0 * Jq);                        // :182  r J
        if (q == 0) { acc[5] = mu; acc[7] = 1.0; }              // terms added, samples ok
    }
    if (q == 0) {
        if (any_viol) acc[6] = 1.0;
    }
    // geometry-only statistics of the own samples, carried by q == 1..4 lanes' slot 5
    if (q == 1 && owned) acc[5] = 1.0;
    if (q == 2 && cls == kClsOog) acc[5] = 1.0;
    if (q == 3 && n < p.n_samples && cls == kClsSkip) acc[5] = 1.0;
    if (q == 4 && n < p.n_samples) acc[5] = 1.0;

    // ---- reduction over the 8 groups of the wave (xor 8, 16, 32), then the 4 waves through LDS
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        double v = acc[e];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        acc[e] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_red[wv][lane][e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        const int qq = tid >> 3, e = tid & 7;
        double v = s_red[0][qq][e];
        for (int w = 1; w < NW; ++w) v += s_red[w][qq][e];
        // row layout: terms [5*q + d]; counters after them
        int slot = -1;
        if (qq < 6 && e < 5) slot = 5 * qq + e;
        else if (qq == 0 && e == 5) slot = kPartTerms;
        else if (qq == 0 && e == 6) slot = kPartViol;
        else if (qq == 0 && e == 7) slot = kPartOk;
        else if (qq == 1 && e == 5) slot = kPartInOwned;
        else if (qq == 2 && e == 5) slot = kPartOog;
        else if (qq == 3 && e == 5) slot = kPartNan;
        else if (qq == 4 && e == 5) slot = kPartSamples;
        if (qq == 7 && e == 7) { slot = kPartWidth - 1; v = fold.tag; }
        if (slot >= 0) store_sc1(&partials[(long long)blockIdx.x * kPartWidth + slot], v);
    }

    // ---- in-launch fan-in (no second launch, no host-side fold): every workgroup has written its row write-through;
    // one lane arrives on the counter of its shard (blockIdx % 8: workgroups b and b + 8 share an XCD, so a shard's
    // arrivals stay on one L2 -- speed only, nothing depends on the placement); the workgroup whose arrival completes a
    // shard folds that shard's rows in row order and arrives on the top counter; the workgroup that completes the top
    // counter adds the shard rows in shard order and hands the result out.  Every sum has a fixed order: the result
    // does not depend on which workgroups happen to arrive last.  Protocol (MI355X_MICROARCH.md, valid forms): sc1
    // stores -> the storing wave's s_waitcnt vmcnt(0) -> ONE lane's device-scope atomic add; the reader is told by the
    // value its own add returned and loads (sc1) only after that.
    __shared__ int s_role;
    __shared__ double s_fold[kTrackBlock / kPartWidth][kPartWidth];
    const unsigned n_wg = gridDim.x;
    const unsigned n_shards = n_wg < (unsigned)kTrackShards ? n_wg : (unsigned)kTrackShards;
    const unsigned shard = blockIdx.x % kTrackShards;
    if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave wrote the row: drained before the arrival
        if (tid == 0) {
            const unsigned in_shard = (n_wg - shard + kTrackShards - 1) / kTrackShards;
            const unsigned old = __hip_atomic_fetch_add(&fold.ctr[32 * shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_role = (old == in_shard - 1u) ? 1 : 0;
        }
    }
    __syncthreads();
    if (s_role == 0) return;

    constexpr int RG = kTrackBlock / kPartWidth;                      // row groups of kPartWidth columns
    const int col = tid % kPartWidth, rg = tid / kPartWidth;
    bool stale = false;
    for (int attempt = 0; attempt < 3; ++attempt) {
        double v = 0.0;
        bool bad = false;
        if (rg < RG) {
            // rows shard, shard + 8, ...: this thread adds every RG-th of them, in order
            constexpr int NF = 12;                                    // loads in flight per thread (714 workgroups: 10 rows per thread)
            double part[NF];
            unsigned r = shard + (unsigned)kTrackShards * (unsigned)rg;
            while (r < n_wg) {
                int nld = 0;
#pragma unroll
                for (int u = 0; u < NF; ++u) {                        // NF loads in flight, summed in row order
                    const unsigned ru = r + (unsigned)(kTrackShards * RG) * (unsigned)u;
                    part[u] = ru < n_wg ? load_sc1(&partials[(long long)ru * kPartWidth + col]) : 0.0;
                    nld += ru < n_wg ? 1 : 0;
                }
#pragma unroll
                for (int u = 0; u < NF; ++u) {
                    if (u < nld) {
                        if (col == kPartWidth - 1) bad |= part[u] != fold.tag;
                        v += part[u];
                    }
                }
                r += (unsigned)(kTrackShards * RG) * (unsigned)NF;
            }
            s_fold[rg][col] = v;
        }
        stale = __syncthreads_or(bad ? 1 : 0) != 0;
        if (!stale) break;
        // a row of another pass: not expected with the protocol above; invalidate this CU's L1 and read again
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    double shard_v = 0.0;
    if (tid < kPartWidth) {
        shard_v = s_fold[0][tid];
        for (int g2 = 1; g2 < RG; ++g2) shard_v += s_fold[g2][tid];
        if (tid == kPartWidth - 1) shard_v = stale ? -1.0 : fold.tag;  // the shard row's own tag
    }
    if (fold.host_shards) {
        // Single-rank hand-off: the (at most 8) shard rows go straight to pinned host memory and the host adds them in
        // shard order -- the second level of the fan-in (another device-scope hand-off: store, drain, atomic, load) is a
        // few hundred host cycles instead of ~2 us on the device.  Every value travels with the pass word in ONE 16-byte
        // store, so no system-scope fence (0.6 us, measured) has to sit between the values and a word behind them: the
        // host takes a value when the word next to it is this pass's.
        if (tid < 64) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2* slot = reinterpret_cast<u64x2*>(fold.host_shards + (size_t)shard * kShardSlotDoubles);
            if (tid < kPartWidth) {
                // The word is mixed with the value's own bits (shard_pair_word): should the 16 bytes ever reach host memory in
                // two pieces -- neither the single global_store_dwordx4 nor an undivided PCIe write is architecturally
                // promised -- an old value next to a new word (or the reverse) does not validate and the host keeps waiting.
                u64x2 pr; pr.x = (unsigned long long)__double_as_longlong(shard_v); pr.y = shard_pair_word(pr.x, fold.word);
                __builtin_nontemporal_store(pr, &slot[tid]);
            }
            if (tid == 0) __hip_atomic_store(&fold.ctr[32 * shard], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next pass
        }
        return;
    }
    if (tid < kPartWidth) store_sc1(&fold.shard_rows[shard * kPartWidth + tid], shard_v);
    __syncthreads();
    if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&fold.ctr[32 * kTrackShards], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_role = (old == n_shards - 1u) ? 2 : 0;
        }
    }
    __syncthreads();
    if (s_role != 2) return;

    // ---- the last shard: add the shard rows in shard order, convert to the result row, hand it out
    __shared__ double s_tot[kPartWidth];
    __shared__ double s_res[kRedWidth];
    stale = false;
    for (int attempt = 0; attempt < 3; ++attempt) {
        bool bad = false;
        if (rg < (int)n_shards && rg < RG) {
            const double v = load_sc1(&fold.shard_rows[rg * kPartWidth + col]);
            if (col == kPartWidth - 1) bad = v != fold.tag;
            s_fold[rg][col] = v;
        }
        stale = __syncthreads_or(bad ? 1 : 0) != 0;
        if (!stale) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    static_assert(kTrackShards <= kTrackBlock / kPartWidth, "one row group per shard in the final sum");
    if (tid < kPartWidth) {
        double v = s_fold[0][tid];
        for (unsigned g2 = 1; g2 < n_shards; ++g2) v += s_fold[g2][tid];
        s_tot[tid] = v;
    }
    __syncthreads();
    if (tid < kRedWidth) {
        double v = 0.0;
        if (tid < 21) {
            // upper triangle, row-major: (a,b) with a <= b  ->  product slot of q = a or q = b
            int a = 0, e = tid;
            while (e >= 6 - a) { e -= 6 - a; ++a; }
            const int b = a + e, d = b - a;
            v = (d <= 3) ? s_tot[5 * a + d] : s_tot[5 * b + (6 - d)];  // (a,b) = (q,(q+d')%6) with q = b, d' = 6-d
        } else if (tid < 27) v = s_tot[5 * (tid - 21) + 4];
        else if (tid == 27) v = s_tot[kPartTerms];
        else if (tid == 28) v = s_tot[kPartViol];
        else if (tid == 29) v = s_tot[kPartOk];
        else if (tid == 30) v = s_tot[kPartInOwned];
        else if (tid == 31) v = s_tot[kPartOog];
        else if (tid == 32) v = s_tot[kPartNan];
        else if (tid == 33) v = s_tot[kPartSamples];
        // a row that stayed stale through two L1 invalidations: the hand-off protocol is broken; poison the term
        // count so that the host refuses the pass instead of solving with an old row
        if (stale && tid == 27) v = __longlong_as_double((long long)kRowPoisonStale);
        if (fold.red_dev && fold.peers.n == 0) fold.red_dev[tid] = v;
        s_res[tid] = v;
    }
    // the counters go back to zero for the next pass (launches of one stream are ordered; nobody else is left in this one)
    if (tid <= kTrackShards) __hip_atomic_store(&fold.ctr[32 * tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // several ranks with a device-side exchange: this rank's row goes to every rank, the sum over ranks comes back
    if (fold.peers.n > 0 && tid < 64) {
        const bool poisoned = s_res[27] != s_res[27];      // a stale fan-in: the NaN travels through every rank's sum
        const bool ok = peer_exchange_row(fold.peers, s_res, kRedAllreduce, tid);
        if (!ok && !poisoned && tid == 27) s_res[27] = __longlong_as_double((long long)kRowPoisonPeerTimeout);
        if (fold.red_dev && tid < kRedWidth) fold.red_dev[tid] = s_res[tid];
    }
    // host hand-off without a stream synchronisation: one wave writes the row to pinned host memory, fences at system
    // scope, then releases the word the host spins on
    if (fold.host_row && tid < 64) {
        if (tid < kRedWidth) fold.host_row[tid] = s_res[tid];
        __threadfence_system();
        if (tid == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(fold.host_row + kRedWidth), fold.word, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// (track_kernel keeps its name and argument list: the stand-alone code object of the library's own queue is found by them)
__global__ __launch_bounds__(kTrackBlock) void track_kernel(TrackParams p, const float2* __restrict__ dw,
                                                             const float4* __restrict__ samples,
                                                             double* __restrict__ partials, TrackFold fold) {
    track_body<false>(p, dw, samples, partials, fold);
}
__global__ __launch_bounds__(kTrackBlock) void track_kernel_cyclic(TrackParams p, const float2* __restrict__ dw,
                                                                    const float4* __restrict__ samples,
                                                                    double* __restrict__ partials, TrackFold fold) {
    track_body<true>(p, dw, samples, partials, fold);
}

// After an in-stream all-reduce (RCCL) of red_dev: hand the reduced row to the host the way track_kernel's last
// workgroup does (pinned memory + system-scope release of the pass number), so the host can poll instead of
// waiting for a stream synchronisation.
__global__ __launch_bounds__(64) void track_publish_kernel(const double* __restrict__ red_dev,
                                                            double* __restrict__ red_host, unsigned long long seq) {
    if (threadIdx.x == 0) {
        for (int e = 0; e < kRedWidth; ++e) red_host[e] = red_dev[e];
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(red_host + kRedWidth), seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_track_publish(hipStream_t s, const double* red_dev, double* red_host, unsigned long long seq) {
    track_publish_kernel<<<dim3(1), dim3(64), 0, s>>>(red_dev, red_host, seq);
    return hipGetLastError();
}

int track_num_blocks(int32_t n_samples) { return (n_samples + kSamplesPerBlock - 1) / kSamplesPerBlock; }
size_t track_partials_doubles(int32_t n_samples) { return ((size_t)track_num_blocks(n_samples) + kTrackShards) * kPartWidth; }

// track_kernel's explicit arguments as a code object lays them out: by-value structs and pointers in declaration order,
// each at its natural alignment (AqlQueue::init checks the total against the code object's kernarg segment)
struct TrackKernarg { TrackParams p; const float2* dw; const float4* samples; double* partials; TrackFold fold; };
static_assert(sizeof(TrackKernarg) % 8 == 0, "kernel arguments are 8-byte aligned");

// One launch per pass: rows, fan-in and result row inside track_kernel.  ctr: track_fold_counter_words() unsigned, zero
// before the first pass (the kernel re-zeroes them); shard rows live behind the per-workgroup rows in `partials`.
hipError_t launch_track_folded(hipStream_t s, const TrackParams& p, const float2* dw, const float4* samples,
                               double* partials, unsigned* ctr, double* red_dev, double* host_row, double* host_shards,
                               unsigned long long word, unsigned long long pass, const PeerExchange* peers,
                               AqlQueue* aql, bool* went_through_queue) {
    const int nb = track_num_blocks(p.n_samples);
    if (nb <= 0) return hipErrorInvalidValue;
    if (peers && (peers->n < 0 || peers->n > kPeerMaxRanks || host_shards)) return hipErrorInvalidValue;
    TrackFold f;
    f.ctr = ctr;
    f.shard_rows = partials + (size_t)nb * kPartWidth;
    f.red_dev = red_dev;
    f.host_row = host_row;
    f.host_shards = host_shards;
    f.word = word;
    f.tag = (double)(pass & 0xFFFFFFFFFFFFull);
    if (peers) f.peers = *peers;
    if (went_through_queue) *went_through_queue = false;
    if (p.g.blk_own > 0) {
        track_kernel_cyclic<<<dim3(nb), dim3(kTrackBlock), 0, s>>>(p, dw, samples, partials, f);
        return hipGetLastError();
    }
    if (aql) {
        TrackKernarg ka{p, dw, samples, partials, f};
        if (aql->submit(&ka, (uint32_t)nb, (uint32_t)kTrackBlock)) { if (went_through_queue) *went_through_queue = true; return hipSuccess; }
    }
    track_kernel<<<dim3(nb), dim3(kTrackBlock), 0, s>>>(p, dw, samples, partials, f);
    return hipGetLastError();
}
size_t track_kernel_explicit_arg_bytes() { return sizeof(TrackKernarg); }
const char* track_kernel_symbol_prefix() { return "_ZN4tsdf12track_kernelE"; }
const char* track_kernel_build_id() { return TSDF_BUILD_ID; }

// (two sets, used alternately by pass parity: with passes coming off two queues a set's re-zeroing store is no longer
// ordered before the NEXT pass's arrivals, only before the one after it)
size_t track_fold_counter_words() { return 32 * (size_t)(kTrackShards + 1); }
int track_num_shards(int32_t n_samples) {
    const int nb = track_num_blocks(n_samples);
    return nb < kTrackShards ? nb : kTrackShards;
}

}  // namespace tsdf
