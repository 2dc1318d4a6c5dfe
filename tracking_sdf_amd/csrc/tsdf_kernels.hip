// tsdf_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the tracking_sdf hot path.
//
//   list_rows_kernel   per k-row frustum interval -> list of 64-voxel work items, sorted by image band
//   integrate_kernel   SDF::update over that list          (reference src/sdf.cpp:224-315)
//   track_kernel       one Gauss-Newton accumulation pass    (reference src/camera_tracking.cpp:146-189,
//                      + get_partial_derivative :246-363, SDF::interpolate_distance sdf.cpp:127-163)
//                      incl. the fixed-order fan-in of the per-workgroup partial normal equations (one launch per pass)
//   pack_kernel        per-frame image packing (xyz|nrm|rgb planes -> 32-byte pixel records + the
//                      tracker's column-major stride-3 sample list, camera_tracking.cpp:162-163)
//   sample_kernel, fill_kernel, split/merge kernels: interpolate_distance batches, constructor fill, host mirrors
//
// Numerics: every operation that decides a result (f64 geometry, f32 interpolation and running
// averages, (int) truncations) is the reference's operation in the reference's order, so this file
// MUST be compiled with -ffp-contract=off (no FMA contraction) and without fast-math.
// Both kernels are HBM/L2-bound byte movers: no MFMA anywhere (there is no dense contraction).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "aql_queue.hpp"
#include "tsdf_device.h"

namespace tsdf {

// ------------------------------------------------------------------------------------------------
// small device helpers

// (int)float / (int)double as x86-64 cvttss2si / cvttsd2si: out-of-range and NaN give INT_MIN
// (the reference relies on this at sdf.cpp:143-145 and :251-252; v_cvt_i32_f32 would saturate).
__device__ __forceinline__ int trunc_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// Eigen 3.2 coefficient-based 3x3 * 3: ((a0*b0 + a1*b1) + a2*b2), see DESIGN.md "evaluation orders".
__device__ __forceinline__ double row3(const double* M, double x, double y, double z) {
    return (M[0] * x + M[1] * y) + M[2] * z;
}

__device__ __forceinline__ bool is_nan(float f) { return f != f; }

// sdf.cpp:294  cosine = fabs(cam_vect.dot(n)) / n.norm(), cam_vect = (0,0,1), in the reference's f64
// evaluation order (Eigen redux: a0*b0 + (a1*b1 + a2*b2)).
__device__ __forceinline__ double pixel_cosine(float nx, float ny, float nz) {
    const double x = (double)nx, y = (double)ny, z = (double)nz;
    return fabs(0.0 * x + (0.0 * y + 1.0 * z)) / sqrt(x * x + (y * y + z * z));
}

// The same value for normals of ordinary size, without the wrappers hipcc puts around sqrt() and '/': its correctly
// rounded cores as they stand in this build's ISA (v_rsq_f64 + the coupled iteration with two residual corrections;
// v_rcp_f64 + two Newton steps, quotient, one residual correction), minus the range scaling (v_div_scale / v_ldexp),
// which only acts outside 2^+-767 / 2^+-1022, and minus the special-case selects (v_div_fixup, the 0 / inf pass-through).
// With x, y finite, fabs(0*x + (0*y + z)) is |z|.  Valid for 2^-200 <= n2 <= 2^200 (the caller tests n2, which also
// rules out NaN and infinite components); bit-identical to pixel_cosine() there -- the fuzz and parity tests compare
// the colour weights of band voxels with the oracle's libm sqrt and division.
__device__ __forceinline__ double pixel_cosine_core(double z, double n2) {
    const double y = __builtin_amdgcn_rsq(n2);
    double g = n2 * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, n2);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, n2);
    g = __builtin_fma(d, h, g);                               // sqrt(n2)
    const double a = __builtin_fabs(z);
    double q = __builtin_amdgcn_rcp(g);
    double e = __builtin_fma(-g, q, 1.0);
    q = __builtin_fma(q, e, q);
    e = __builtin_fma(-g, q, 1.0);
    q = __builtin_fma(q, e, q);
    const double t = a * q;
    const double res = __builtin_fma(-g, t, a);
    return __builtin_fma(res, q, t);
}

// ------------------------------------------------------------------------------------------------
// volume fill: D = width+height+depth, W = 0, Color_W = 0, R = G = B = 0.4f   (sdf.cpp:28-34)

__global__ __launch_bounds__(256) void fill_kernel(float2* __restrict__ dw, float4* __restrict__ crgb,
                                                    long long n, float d0) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        dw[i] = make_float2(d0, 0.0f);
        if (crgb) crgb[i] = make_float4(0.0f, 0.4f, 0.4f, 0.4f);
    }
}

hipError_t launch_fill(hipStream_t s, const Grid& g, float2* dw, float4* crgb, float d0) {
    const long long n = (long long)(g.xe - g.xs) * g.m * g.m;
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dw, crgb, n, d0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// frame packing.  Per pixel one record {Px,Py,Pz, rgb bits} {Nx,Ny,Nz, (float)cosine} (32 bytes; 24 bytes {P,N} for a
// volume without colour; kPixelRecordBytes per pixel allocated).  One record
// per projected voxel instead of scattered plane reads.  Records are stored row-major
// or column-major (pix_su / pix_sv), whichever makes the pixels hit by 64 consecutive k of one voxel
// row neighbours in memory: a k-row projects to a near-vertical image line for an upright camera,
// and with row-major records every lane of the gather then pulls its own 128-byte line through L2
// (measured: 0.6 ms of L2->L1 line traffic per 512^3 frame, the v1 bottleneck).  The tracker's sample list is
// written in the reference's visiting order: columns outer, rows inner, both with `stride`.

__device__ __forceinline__ void pack_tile(const PackArgs& a_, int tile) {
    const float* __restrict__ xyz = a_.xyz; const float* __restrict__ nrm = a_.nrm; const uint8_t* __restrict__ rgb = a_.rgb;
    const int width = a_.width, height = a_.height, stride = a_.stride, pix_su = a_.pix_su, pix_sv = a_.pix_sv;
    float4* __restrict__ pn = a_.pn; float4* __restrict__ samples = a_.samples;
    const int ncols = a_.ncols, nrows = a_.nrows, color_layout = a_.color_layout;
    // 16 x 16 pixel tiles (256 threads); consecutive threads follow the direction in which the records are contiguous,
    // so a wavefront writes four runs of 512 bytes whichever layout is chosen (the plane reads of a tile stay within
    // a few cache lines per image row either way)
    const int tiles_x = (width + 15) >> 4;
    const int tx0 = (tile % tiles_x) << 4, ty0 = (tile / tiles_x) << 4;
    const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
    const int col = tx0 + (pix_sv == 1 ? a : b), row = ty0 + (pix_sv == 1 ? b : a);
    if (col >= width || row >= height) return;
    const int pix = row * width + col;
    const float qnan = __int_as_float(0x7fc00000);
    const float px = xyz[3 * pix + 0], py = xyz[3 * pix + 1], pz = xyz[3 * pix + 2];
    float nx = qnan, ny = qnan, nz = qnan;
    if (nrm) { nx = nrm[3 * pix + 0]; ny = nrm[3 * pix + 1]; nz = nrm[3 * pix + 2]; }
    unsigned c = 0;
    if (rgb) c = (unsigned)rgb[3 * pix + 0] | ((unsigned)rgb[3 * pix + 1] << 8) | ((unsigned)rgb[3 * pix + 2] << 16);
    const long long rec = (long long)col * pix_su + (long long)row * pix_sv;   // row- or column-major records
    if (color_layout) {
        // with colour: 32-byte records {Px,Py,Pz, rgb bits} {Nx,Ny,Nz, (float)cosine}.  sdf.cpp:294: cosine =
        // |cam_vect . n| / |n| depends on the pixel only; its f32 rounding rides in the record: for the common weight
        // w_new == 1 the colour weight (float)(w_new * cosine) is exactly that ...
        // (the voxels of the exp() band, whose weight is not 1, recompute the f64 cosine from the normal)
        const double cosine = pixel_cosine(nx, ny, nz);
        pn[2 * rec + 0] = make_float4(px, py, pz, __uint_as_float(c));
        pn[2 * rec + 1] = make_float4(nx, ny, nz, (float)cosine);
        // ... and the f64 value itself goes to a plane behind the records (8 bytes per pixel, same record index): the
        // queue kernel gathers it for DENSE batches of exp()-band voxels only (one 8-byte gather per 64 band voxels)
        if (color_layout == 2) reinterpret_cast<double*>(pn + 2 * (size_t)width * height)[rec] = cosine;
    } else {
        // without colour: 24-byte records {Px,Py,Pz, Nx,Ny,Nz} (a quarter fewer cache lines per gathered pixel run)
        float* const r6 = reinterpret_cast<float*>(pn) + rec * 6;
        r6[0] = px; r6[1] = py; r6[2] = pz; r6[3] = nx; r6[4] = ny; r6[5] = nz;
    }
    if (col % stride == 0 && row % stride == 0) {
        const int ci = col / stride, rj = row / stride;
        if (samples && ci < ncols && rj < nrows) samples[ci * nrows + rj] = make_float4(px, py, pz, 0.0f);
    }
}

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) { pack_tile(a, (int)blockIdx.x); }

static int pack_tiles(const PackArgs& a) { return ((a.width + 15) >> 4) * ((a.height + 15) >> 4); }

hipError_t launch_pack(hipStream_t s, const PackArgs& a) {
    pack_kernel<<<dim3(pack_tiles(a)), dim3(256), 0, s>>>(a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// TSDF integration = SDF::update (reference src/sdf.cpp:224-315), in two launches.
//
// The reference visits all m^3 voxels and rejects most of them (behind the camera / outside the
// image / NaN pixel / d > delta); rejected voxels cost no HBM traffic.  Measured on MI355X (round 1):
// a kernel that walks every voxel is neither ALU- nor HBM-bound but LATENCY-bound -- VALU busy 16 %,
// waves parked 65 % of their life, ~1 resident wave per SIMD on average -- because the few wavefronts
// that own in-frustum voxels run long serial chains (pixel gather -> {D,W} read -> write) while the
// rest of the chip has nothing to do.  So the work is first compacted, then spread evenly:
//
//   list_rows_kernel   one thread per k-row (fixed i,j; k = 0..m-1).  A row is a straight segment in
//                      camera space, pc(k) = Q0 + k Q1, and every frustum test (z >= 0, u > -1, u < W,
//                      v > -1, v < H) is affine in k, so the k that can pass form ONE interval, found
//                      with five reciprocals per row.  The interval is widened by a voxel per side and
//                      its 64-voxel chunks go to the work list, into the region of the image band the row
//                      projects to.  Every listed voxel still runs the reference's exact tests, so the
//                      cull never changes a result.
//   integrate_kernel   every XCD takes a contiguous part of the list, its persistent workgroups walk it together; one item = 64 consecutive k of
//                      one row = one 512-byte {D,W} segment (+1 KiB colour): perfectly coalesced RMW.
//                      The row's share of rot_inv * g (its first two terms, identical for all k) travels in the
//                      item descriptor (one scalar 32-byte load per item), the third term comes from a table of
//                      the m values of k in LDS: two f64 adds per camera coordinate and voxel.
//
// Algorithmic traffic: 16 B (48 B with colour) per *updated* voxel + the 32-byte pixel records.

#ifndef TSDF_CLIP_BLOCK
#define TSDF_CLIP_BLOCK 256
#endif
constexpr int kClipBlock = TSDF_CLIP_BLOCK;      // rows (threads) per workgroup of list_rows_kernel
#ifndef TSDF_INTEGRATE_MIN_WAVES
#define TSDF_INTEGRATE_MIN_WAVES 5   // waves per SIMD the register allocator must leave room for (<= 96 VGPRs; the peeled pipeline took 108 at 4)
#endif

struct IntegrateTiling {
    long long n_rows;    // (xe-xs) * m
    int log2m;           // >= 0 when m is a power of two
    int clip;            // 1 = K has the usual last row (0,0,k22>0): row clipping is valid
    int k_std;           // 1 = K = [[fx,0,cx],[0,fy,cy],[0,0,1]] exactly: zero terms can be dropped
    int fastq;           // 1 = pixel coordinates by fixed-point reciprocal multiplies (integrate_kernel), 0 = always divide
};

// interval of k (real-valued) on which a + k*b > 0, intersected into [lo, hi].  The crossing -a/b only has to be
// good to a fraction of a voxel (the interval gets a whole voxel of slack per side and every voxel inside still
// runs the reference's exact test), so it is a refined reciprocal times -a, not an IEEE division.
__device__ __forceinline__ double clip_crossing(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    return -a * r;
}
__device__ __forceinline__ void clip_affine(double a, double b, double& lo, double& hi) {
    if (b > 0.0) { const double t = clip_crossing(a, b); if (t > lo) lo = t; }
    else if (b < 0.0) { const double t = clip_crossing(a, b); if (t < hi) hi = t; }
    else if (a < -1.0e-9) { lo = 1.0; hi = 0.0; }   // row parallel to this plane and clearly outside it
    // (b == 0 and a within rounding of 0: leave it to the exact per-voxel test)
}

// Work-list bookkeeping of one launch (a "set", words):
//   [kSetCur + b]     cursor of image band b: items of band b handed out so far = the band's item count at the end
//   [kSetFirstOvf + b] smallest cursor value at which a group of band b did not fit its region any more (~0: all fitted)
//   [kSetCap + b]     capacity of band b's region in the list          } prepared by the PREVIOUS launch's
//   [kSetBase + b]    first list entry of band b's region (b = 0..kBins: [kBins] = end of the regions)  } integrate_kernel
//   [kSetOvf]         items in the overflow region (list entries [ovf_base, ...), ovf_base = integrate_band_region_entries())
// One pass builds the band-sorted list (list_rows_kernel): the band regions are sized from the band counts of the
// previous frame (+25 % + 128 entries) -- consecutive frames see nearly the same image -- and whatever does not fit goes
// to the overflow region behind them, which is integrated like any other part of the list, only without the band's
// locality.  The first launch after creation (all capacities 0) puts everything there.  Two sets are used
// alternately: a launch's integrate_kernel prepares the set of the NEXT launch (nobody else touches it meanwhile).
#ifndef TSDF_BANDS
#define TSDF_BANDS 64
#endif
constexpr int kBins = TSDF_BANDS;
static_assert(kBins >= 8 && kBins <= 4096 && (kBins & (kBins - 1)) == 0, "bands: a power of two that fits the row word");
enum { kSetCur = 0, kSetFirstOvf = kBins, kSetCap = 2 * kBins, kSetBase = 3 * kBins, kSetOvf = 4 * kBins + 1 };
constexpr int kBinSetWords = 4 * kBins + 2;

// Shares of the eight XCDs in the band-sorted work list, adjusted from launch to launch.  Items differ in cost (live
// lanes, lines touched) and an XCD's band keeps its character from frame to frame; with equal item counts the slowest
// XCD finished 10-15 % after the fastest.  Feedback block behind the two bookkeeping sets: words [0..8] = share
// boundaries as fractions of the list in 2^-24 units (0 .. 2^24), then (8-byte aligned) eight 64-bit sums of what the
// first wavefronts of the XCD's workgroups measured for their item loops in the LAST launch (s_memrealtime ticks).
// list_rows_kernel (its last block) turns them into the next boundaries: share ~ items per tick, half-way damped, each
// share kept within [1/16, 1/4].  Only the schedule depends on it -- every voxel belongs to exactly one item.
constexpr int kFbWords = 32;
constexpr int kFbTicksWord = 16;
constexpr unsigned kFbOne = 1u << 24;
#ifndef TSDF_XCD_FEEDBACK
#define TSDF_XCD_FEEDBACK 1
#endif
__device__ __forceinline__ void update_xcd_shares(unsigned* fb) {
    unsigned long long* ticks = reinterpret_cast<unsigned long long*>(fb + kFbTicksWord);
    if (fb[8] != kFbOne) {                                   // first launch: equal shares
        for (int x = 0; x <= 8; ++x) fb[x] = (unsigned)x * (kFbOne / 8u);
    } else if (TSDF_XCD_FEEDBACK) {
        double rate[8], sum = 0.0;
        bool ok = true;
        for (int x = 0; x < 8; ++x) {
            const double share = (double)(fb[x + 1] - fb[x]);
            ok &= ticks[x] != 0ull;
            rate[x] = ok ? share / (double)ticks[x] : 0.0;
            sum += rate[x];
        }
        if (ok && sum > 0.0) {
            double sh[8], tot = 0.0;
            for (int x = 0; x < 8; ++x) {
                const double target = rate[x] / sum, old = (double)(fb[x + 1] - fb[x]) / (double)kFbOne;
                double v = 0.5 * old + 0.5 * target;
                v = v < 1.0 / 16.0 ? 1.0 / 16.0 : (v > 0.25 ? 0.25 : v);
                sh[x] = v; tot += v;
            }
            double run = 0.0;
            for (int x = 0; x < 8; ++x) { fb[x] = (unsigned)(run / tot * (double)kFbOne); run += sh[x]; }
            fb[0] = 0u; fb[8] = kFbOne;
        }
    }
    for (int x = 0; x < 8; ++x) ticks[x] = 0ull;
}

// One work item as integrate_kernel reads it, with ONE scalar load: the item code and the row's share of rot_inv * g
// (its first two terms in Eigen's order ((r0*gx + r1*gy) + r2*gz), identical for every k of the row).
struct __attribute__((aligned(32))) ItemDesc {
    unsigned code;          // row << 6 | chunk   (row = il * m + j, chunk = k / 64)
    unsigned pad;
    double s0, s1, s2;
};
static_assert(sizeof(ItemDesc) == 32, "one s_load_dwordx8 per item");

// The list of work items, sorted by image band, in ONE pass (round 3; rounds 1-2 clipped the rows in one kernel and
// scattered their items in a second one, because the band counts had to be complete before the first item could be
// placed: two latency-bound kernels of 7 and 10 us).  One thread per voxel row clips it against the frustum; the
// workgroup's rows are counted per band in LDS, ONE returning atomic per band and workgroup reserves their place in the
// band's region -- or, when the region is full, in the overflow region -- and all threads write the descriptors.
// integrate_kernel hands each XCD one contiguous part of the list = a band of the image whose pixel records (about
// 1.2 MB) then live in that XCD's L2: with the list in row order every XCD gathered from the whole image, and 63 % of
// the launch's fabric reads were pixel records fetched again and again (296 MB for a 9.8 MB image).  The order inside a
// band is whatever the atomics give -- every voxel belongs to exactly one item, so no result depends on it.
//
// Deferred frame packing (round 4): the workgroups behind the list's own (blockIdx >= list_blocks) write the frame's pixel
// records (pack_tile) -- for frames handed over in device memory the tracker reads its samples straight from the xyz
// plane, so nothing needs the records before integrate_kernel and the packing hides under this kernel's latency
// chain (one thread per row, three barriers, two atomic round trips) instead of being a launch of its own.
static_assert(kClipBlock == 256, "pack_tile works on 256-thread workgroups");
__global__ __launch_bounds__(kClipBlock) void list_rows_kernel(IntegrateParams p, IntegrateTiling tl,
                                                                unsigned* __restrict__ set, ItemDesc* __restrict__ list,
                                                                unsigned ovf_base, unsigned* __restrict__ xcd_fb,
                                                                unsigned list_blocks, PackArgs pack) {
    if (blockIdx.x >= list_blocks) { pack_tile(pack, (int)(blockIdx.x - list_blocks)); return; }
    const int m = p.g.m;
    const int tid = threadIdx.x;
    const long long row = (long long)blockIdx.x * kClipBlock + tid;
    if (blockIdx.x == list_blocks - 1 && tid == 0) update_xcd_shares(xcd_fb);   // (the last launch's integrate kernel is done: same stream)
    __shared__ unsigned s_wg[kBins], s_dest[kBins];
    for (int t = tid; t < kBins; t += kClipBlock) s_wg[t] = 0u;
    __syncthreads();
    int c0 = 0, n = 0, bin = 0;
    if (row < tl.n_rows) {
        int il, j;
        if (tl.log2m >= 0) { il = (int)(row >> tl.log2m); j = (int)(row & (m - 1)); }
        else { il = (int)(row / m); j = (int)(row - (long long)il * m); }
        const double cw = (double)p.g.cell_w, ch = (double)p.g.cell_h, cd = (double)p.g.cell_d;
        // get_global_coordinates, sdf.h:153-157: (extent/(float)m) * (i + 0.5) + origin
        const double gx = cw * ((double)(il + p.g.xs) + 0.5) + p.g.origin[0];
        const double gy = ch * ((double)j + 0.5) + p.g.origin[1];
        // first two terms of rot_inv * g in Eigen's order ((r0*gx + r1*gy) + r2*gz): the same for every k
        double S[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) S[a] = p.rot_inv[3 * a] * gx + p.rot_inv[3 * a + 1] * gy;
        int klo = 0, khi = m - 1;
        if (tl.clip) {
            const double gz0 = cd * 0.5 + p.g.origin[2];               // k = 0
            double Q0[3], Q1[3];                                       // pc(k) = Q0 + k Q1
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                Q0[a] = S[a] + p.rot_inv[3 * a + 2] * gz0 + p.rot_inv_trans[a];
                Q1[a] = p.rot_inv[3 * a + 2] * cd;
            }
            // ij = K pc, with K's last row (0,0,k22): ij2 = k22 * pcz has the sign of pcz
            const double a0 = row3(&p.K[0], Q0[0], Q0[1], Q0[2]), b0 = row3(&p.K[0], Q1[0], Q1[1], Q1[2]);
            const double a1 = row3(&p.K[3], Q0[0], Q0[1], Q0[2]), b1 = row3(&p.K[3], Q1[0], Q1[1], Q1[2]);
            const double a2 = p.K[8] * Q0[2], b2 = p.K[8] * Q1[2];
            double lo = -1.0, hi = (double)m;
            clip_affine(a2, b2, lo, hi);                                                   // pcz >= 0
            clip_affine(a0 + a2, b0 + b2, lo, hi);                                         // u > -1
            clip_affine((double)p.width * a2 - a0, (double)p.width * b2 - b0, lo, hi);     // u < W
            clip_affine(a1 + a2, b1 + b2, lo, hi);                                         // v > -1
            clip_affine((double)p.height * a2 - a1, (double)p.height * b2 - b1, lo, hi);   // v < H
            if (!(lo <= hi + 1.0e-6)) { klo = 1; khi = 0; }            // empty (or NaN): nothing can pass
            else {
                const double l2 = floor(lo) - 1.0, h2 = ceil(hi) + 1.0;   // one voxel of slack per side
                klo = l2 < 0.0 ? 0 : (l2 > (double)(m - 1) ? m : (int)l2);
                khi = h2 > (double)(m - 1) ? m - 1 : (h2 < 0.0 ? -1 : (int)h2);
                // Image band of the row = where the middle of its interval projects, along the image axis in which
                // the pixel records are NOT contiguous (columns for column-major records).  Only ordering depends on
                // it (which XCD integrates the row, next to which other rows), never a result.
                const double km = 0.5 * ((double)klo + (double)khi);
                const double den = a2 + km * b2;
                const double coord = p.pix_sv == 1 ? (a0 + km * b0) / den : (a1 + km * b1) / den;
                const double ext = p.pix_sv == 1 ? (double)p.width : (double)p.height;
                const double fb = coord * ((double)kBins / ext);
                bin = fb >= 0.0 ? (fb < (double)(kBins - 1) ? (int)fb : kBins - 1) : 0;       // NaN -> 0
            }
        }
        if (klo <= khi) { c0 = klo >> 6; n = (khi >> 6) - c0 + 1; }
    }
#ifdef TSDF_LIST_ABLATION
    if (p.debug & 0x40000) return;                   // timing experiment: the clip alone
#endif
    unsigned rank = 0u;
    if (n) rank = atomicAdd(&s_wg[bin], (unsigned)n);
    __syncthreads();
    // one returning atomic per band with items: the group's place in the band's region, or in the overflow region
    for (int t = tid; t < kBins; t += kClipBlock) {
        const unsigned cnt = s_wg[t];
        if (cnt) {
#ifdef TSDF_LIST_ABLATION
            if (p.debug & 0x10000) { s_dest[t] = ovf_base; continue; }      // timing experiment: no global atomics (the list stays empty)
#endif
            const unsigned at = atomicAdd(&set[kSetCur + t], cnt);
            unsigned dest;
            if (at + cnt <= set[kSetCap + t]) dest = set[kSetBase + t] + at;
            else {
                atomicMin(&set[kSetFirstOvf + t], at);       // the band's region ends being valid here
                dest = ovf_base + atomicAdd(&set[kSetOvf], cnt);
            }
            s_dest[t] = dest;
        }
    }
    __syncthreads();
    // The items are written by ALL threads of the workgroup, one 32-byte descriptor each per round: most rows of a
    // workgroup have no item and a few have up to m/64, so a loop over the own row's chunks left one lane of a
    // wavefront writing while the others waited (11.4 us per launch at 512^3; 6-7 us spread out).
    __shared__ unsigned s_at[kClipBlock], s_n[kClipBlock], s_c0[kClipBlock], s_pre[kClipBlock + 1];
    __shared__ unsigned s_wave[kClipBlock / 64];
    s_at[tid] = n ? s_dest[bin] + rank : 0u;
    s_n[tid] = (unsigned)n;
    s_c0[tid] = (unsigned)c0;
    {
        unsigned incl = (unsigned)n;                        // exclusive scan of the rows' item counts over the workgroup
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off);
            if ((tid & 63) >= off) incl += t;
        }
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        unsigned before = 0u;
#pragma unroll
        for (int w = 0; w < kClipBlock / 64; ++w) if (w < (tid >> 6)) before += s_wave[w];
        s_pre[tid + 1] = before + incl;
        if (tid == 0) s_pre[0] = 0u;
    }
    __syncthreads();
    unsigned total = s_pre[kClipBlock];
#ifdef TSDF_LIST_ABLATION
    if (p.debug & 0x20000) total = 0u;               // timing experiment: no descriptors written
#endif
    for (unsigned e = tid; e < total; e += kClipBlock) {
        // the row of item e: the last r with s_pre[r] <= e
        unsigned lo = 0u, hi = kClipBlock;
        while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (s_pre[mid] <= e) lo = mid; else hi = mid; }
        const unsigned r = lo, q = e - s_pre[r];
        const long long rrow = (long long)blockIdx.x * kClipBlock + r;
        int il, jr;
        if (tl.log2m >= 0) { il = (int)(rrow >> tl.log2m); jr = (int)(rrow & (m - 1)); }
        else { il = (int)(rrow / m); jr = (int)(rrow - (long long)il * m); }
        // get_global_coordinates, sdf.h:153-157: (extent/(float)m) * (i + 0.5) + origin
        const double gx = (double)p.g.cell_w * ((double)(il + p.g.xs) + 0.5) + p.g.origin[0];
        const double gy = (double)p.g.cell_h * ((double)jr + 0.5) + p.g.origin[1];
        ItemDesc d;
        d.pad = 0u;
        d.s0 = p.rot_inv[0] * gx + p.rot_inv[1] * gy;
        d.s1 = p.rot_inv[3] * gx + p.rot_inv[4] * gy;
        d.s2 = p.rot_inv[6] * gx + p.rot_inv[7] * gy;
        d.code = ((unsigned)rrow << 6) | (s_c0[r] + q);
        list[s_at[r] + q] = d;
    }
}

// exp(x) for the weight of sdf.cpp:278.  x = -(d-eps)^2/2 lies in [-(delta-eps)^2/2, 0] = [-0.0378, 0] with the
// reference's delta and epsilon; for |x| <= 0.04 (decided on the host: template flag EXPPOLY) the degree-8 Taylor polynomial
// in f64 (fused multiply-adds: this approximates the exact function, it does not mimic reference roundings) has a
// truncation error below 0.04^9/9! = 7e-19 relative, i.e. it is as close to the true value as glibc's / ocml's exp
// (< 1 ulp of f64 = 1.1e-16) and agrees with them after the reference's f64 -> f32 narrowing except for values within
// ~1e-16 (relative) of an f32 rounding boundary.  Larger |x| (non-default delta) use the library exp.
// v_fma_f64 spelled out: hipcc turns a Horner step with a constant addend into v_mov_b64 + v_fmac_f64.
__device__ __forceinline__ double fma3(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ double exp_taylor8(double x) {
    double r = fma3(x, 1.0 / 40320.0, 1.0 / 5040.0);
    r = fma3(r, x, 1.0 / 720.0);
    r = fma3(r, x, 1.0 / 120.0);
    r = fma3(r, x, 1.0 / 24.0);
    r = fma3(r, x, 1.0 / 6.0);
    r = __builtin_fma(r, x, 0.5);
    r = __builtin_fma(r, x, 1.0);
    r = __builtin_fma(r, x, 1.0);
    return r;
}

// ---- integrate_kernel ---------------------------------------------------------------------------------------------
//
// Round 3: the kernel is bound by VECTOR-INSTRUCTION ISSUE, not by memory (profiles/r03_integrate_ablation.json: with
// every access redirected to cache-resident addresses the round-2 kernel lost 8 us of 122; one wave-instruction
// costs ~0.33 us of launch time whatever its type), so this version is written for the smallest number of vector
// instructions per 64-voxel item that still performs the reference's operations bit for bit:
//   * everything that is the same for the 64 lanes of an item lives in SGPRs: the item descriptor arrives by one
//     scalar load, addresses are scalar bases + a loop-invariant per-lane offset, predicates stay lane masks and
//     are counted with s_bcnt1;
//   * the k-dependent products rot_inv[.,2] * gz(k) come from a table in LDS built once per workgroup (they are
//     the same for every row);
//   * the two projective quotients u = ij0/ij2, v = ij1/ij2 are only needed through (int)u, (int)v and the range
//     tests, so they are computed as ij * (refined f32 reciprocal) in 2^-20 pixel fixed point (error < 2^-10 of a
//     unit, see fast_quotients) and the wavefront falls back to the reference's two f64 divisions whenever a lane
//     lands within 2 units of an integer (about once in 10^4 wavefronts);
//   * volume reads and stores are raw buffer operations on a 512-byte / 1-KiB descriptor of the item's segment:
//     dead lanes carry an out-of-range offset and touch no memory -- no branch, no EXEC juggling, exact s_waitcnt counts;
//   * the four f32 divisions of the running averages (one for D, three for the colour) run two at a time as packed
//     f32 operations with the division's own FMA sequence (exactly the instruction sequence hipcc emits for
//     a correctly rounded `/`, minus the range scaling, which a guard proves unnecessary or else takes the `/` path);
//   * the pre-rounded (float)cosine of the colour weight rides in the pixel record (weight-1 voxels use it as it is);
//     a wavefront with lanes in the exp() band recomputes the f64 cosine from the record's normal for those lanes
//     (a per-pixel plane of f64 cosines gathered per item was measured twice and lost both times: one more
//     vector-memory instruction per item; integrate_queue_kernel gathers it once per 64 band voxels instead).

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kFixShift = 20;                        // pixel coordinates in 2^-20 units
constexpr unsigned kFixOne = 1u << kFixShift;
constexpr int kMaxFastDim = 2047;                    // (dim + 1) << 20 must fit 32 bits
constexpr unsigned kDroppedOffset = 0x7fffffffu;     // beyond every buffer: the lane loads zeros / stores nothing
constexpr int kRsrcWord3 = 0x00020000;               // raw buffer, 32-bit data format (gfx9 family)

// Per-pixel data of a frame, written by pack_kernel into ONE buffer of kPixelBufferBytes per pixel (record index
// rec = col*pix_su + row*pix_sv):  with colour  [0, 32 npix) records {Px,Py,Pz,rgb}{Nx,Ny,Nz,(float)cosine}, then
// [32 npix, 40 npix) the f64 cosines (written only for integrate_queue_kernel);  without colour  [0, 24 npix) records
// {Px,Py,Pz,Nx,Ny,Nz}.
static_assert(kPixelRecordBytes == 32 && kPixelBufferBytes == 40, "pixel records + f64 cosine plane");

// v_cvt_i32_f64 as the hardware does it (saturating, NaN -> 0); a C cast of an out-of-range value is undefined
__device__ __forceinline__ int cvt_i32_f64_sat(double x) {
    int r;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// lane mask of a per-lane condition (v_cmp writes it straight into an SGPR pair)
__device__ __forceinline__ unsigned long long lanes(bool c) { return __builtin_amdgcn_ballot_w64(c); }

// lanes whose value is subnormal (v_cmp_class_f32 with the two subnormal classes), straight into an SGPR pair:
// through __builtin_amdgcn_classf + ballot hipcc makes a 0/1 VGPR of it first
__device__ __forceinline__ unsigned long long lanes_subnormal(float x) {
    unsigned long long m;
    const unsigned cls = (1u << 4) | (1u << 7);
    asm("v_cmp_class_f32 %0, %1, %2" : "=s"(m) : "v"(x), "v"(cls));
    return m;
}

// per-lane select by a wave-uniform lane mask held in an SGPR pair: bit of the lane set ? a : b.  (A bool that
// crosses a loop iteration becomes a 0/1 VGPR + v_and + v_cmp in hipcc's hands; masks carried as 64-bit scalars do not.)
__device__ __forceinline__ unsigned select_by_mask(unsigned long long mask, unsigned a, unsigned b) {
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
// the same for a mask that IS wave-uniform but that hipcc's uniformity analysis may have given up on (the "s" constraint
// would then receive a vector register pair): readfirstlane, folded away when the mask already sits in scalar registers
__device__ __forceinline__ unsigned select_by_uniform_mask(unsigned long long mask, unsigned a, unsigned b) {
    const unsigned long long m = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mask) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mask >> 32)) << 32);
    return select_by_mask(m, a, b);
}

// What both integrate kernels need to turn an item into pixel indices: constants of the launch in registers.
struct ProjConst {
    double Ks[6];            // K rows 0 and 1 times 2^20 (exact): ij0, ij1 come out in 2^-20 pixel units
    double K2[3];            // K row 2
    double t[3];             // rot_inv_trans
    double r2[3];            // rot_inv[2], [5], [8]   (only without the LDS table)
    double oz, cd;
    unsigned lim_u, lim_w;
    unsigned su, sv;         // record index of pixel (col,row) = col*su + row*sv
    unsigned last_chunk;
    unsigned long long tail_mask;
    unsigned tab_lane;       // lane * 24: byte offset of the lane's entry in a chunk of the k table
    float width_f, height_f;
    int width, height;
    bool fastq;
};

__device__ __forceinline__ void make_proj_const(const IntegrateParams& p, const IntegrateTiling& tl, int lane, ProjConst& c) {
    const double fix = (double)kFixOne;
#pragma unroll
    for (int a = 0; a < 6; ++a) c.Ks[a] = p.K[a] * fix;
#pragma unroll
    for (int a = 0; a < 3; ++a) { c.K2[a] = p.K[6 + a]; c.t[a] = p.rot_inv_trans[a]; c.r2[a] = p.rot_inv[3 * a + 2]; }
    c.oz = p.g.origin[2]; c.cd = (double)p.g.cell_d;
    c.lim_u = ((unsigned)p.width + 1u) * kFixOne + 2u; c.lim_w = ((unsigned)p.height + 1u) * kFixOne + 2u;
    c.su = (unsigned)p.pix_su; c.sv = (unsigned)p.pix_sv;
    // lanes of the last chunk that lie inside the grid when m is not a multiple of 64 (scalar select per item)
    c.last_chunk = (unsigned)(p.g.m - 1) >> 6;
    c.tail_mask = (p.g.m & 63) ? ((1ull << (p.g.m & 63)) - 1ull) : ~0ull;
    c.tab_lane = (unsigned)lane * 24u;
    c.width = p.width; c.height = p.height;
    c.fastq = tl.fastq != 0;
}

// the k table in LDS: {rot_inv[2], rot_inv[5], rot_inv[8]} * gz(k), k = 0..m-1 (the same for every voxel row)
__device__ __forceinline__ void build_k_table(const IntegrateParams& p, double* s_tab, int tid, int nthreads) {
    const double oz = p.g.origin[2], cd = (double)p.g.cell_d;
    for (int k = tid; k < p.g.m; k += nthreads) {
        // get_global_coordinates (sdf.h:153-157): (extent/(float)m) * (k + 0.5) + origin; third term of rot_inv * g
        const double gz = cd * ((double)k + 0.5) + oz;
        s_tab[3 * k + 0] = p.rot_inv[2] * gz;
        s_tab[3 * k + 1] = p.rot_inv[5] * gz;
        s_tab[3 * k + 2] = p.rot_inv[8] * gz;
    }
    __syncthreads();
}

// One item -> camera-frame voxel centres, the lanes that can be updated as far as geometry goes (sdf.cpp:244-256), and
// the lanes' pixels as a BIASED record index pixb = (col+1)*su + (row+1)*sv (callers shift their plane bases).
template <bool KSTD, bool KTAB>
__device__ __forceinline__ void project_item(const ProjConst& c, const ItemDesc& ds, const double* s_tab, int lane,
                                             double& pcx, double& pcy, double& pcz, unsigned long long& okm, unsigned& pixb) {
    const unsigned chunk = ds.code & 63u;
    double a0, a1, a2;
    if (KTAB) {
        const double* t = reinterpret_cast<const double*>(reinterpret_cast<const char*>(s_tab) + (c.tab_lane + chunk * (64u * 24u)));
        a0 = t[0]; a1 = t[1]; a2 = t[2];
    } else {
        const double gz = c.cd * ((double)((int)(chunk * 64u) + lane) + 0.5) + c.oz;
        a0 = c.r2[0] * gz; a1 = c.r2[1] * gz; a2 = c.r2[2] * gz;
    }
    // get_global_coordinates (sdf.h:153-157) + project_world_to_camera (camera_tracking.cpp:51-54)
    pcx = (ds.s0 + a0) + c.t[0];
    pcy = (ds.s1 + a1) + c.t[1];
    pcz = (ds.s2 + a2) + c.t[2];
    okm = lanes(!(pcz < 0)) & (chunk == c.last_chunk ? c.tail_mask : ~0ull);              // sdf.cpp:247-249
    // project_camera_to_image_plane, camera_tracking.cpp:40-47, rows 0 and 1 scaled by 2^20 (exact).  With
    // K = [[fx,0,cx],[0,fy,cy],[0,0,1]] the dropped terms are +-0 products: (fx*x + 0*y) + cx*z == fx*x + cx*z
    // and (0*x + 0*y) + 1*z == z bit for bit (up to the sign of a zero, which no later step can observe).
    double ij0, ij1, ij2;
    if (KSTD) {
        ij0 = c.Ks[0] * pcx + c.Ks[2] * pcz;
        ij1 = c.Ks[4] * pcy + c.Ks[5] * pcz;
        ij2 = pcz;
    } else {
        ij0 = row3(&c.Ks[0], pcx, pcy, pcz);
        ij1 = row3(&c.Ks[3], pcx, pcy, pcz);
        ij2 = row3(&c.K2[0], pcx, pcy, pcz);
    }
    // (int)(ij0/ij2), (int)(ij1/ij2) and the range tests of sdf.cpp:250-256 without dividing: ij2's reciprocal
    // from v_rcp_f32 (1 ulp) and one Newton step in f64 is good to 2^-43, so q = ij * rd is within
    // |q| 2^-42 <= 2^-11 units (|q| < 2^31 units) of the true quotient, and so is the reference's rounded
    // quotient (2^-53 relative).  A lane whose q lies within 2 units of a multiple of 2^20 (an integer pixel
    // coordinate: truncation and both range tests switch only there), or whose ij2 is not a plain positive
    // number, sends the wavefront through the reference's divisions.
    const float zf = (float)ij2;
    double rd = (double)__builtin_amdgcn_rcpf(zf);
    rd = __builtin_fma(__builtin_fma(-ij2, rd, 1.0), rd, rd);
    const unsigned tu = (unsigned)cvt_i32_f64_sat(ij0 * rd) + (kFixOne + 2u);
    const unsigned tw = (unsigned)cvt_i32_f64_sat(ij1 * rd) + (kFixOne + 2u);
    unsigned long long inrm = lanes(tu < c.lim_u) & lanes(tw < c.lim_w);
    unsigned iu1 = max(tu >> kFixShift, 1u), iw1 = max(tw >> kFixShift, 1u);              // pixel column + 1, row + 1
    const unsigned long long doubtm =
        okm & (lanes(!(zf > 1.0e-6f)) | lanes(min(tu & (kFixOne - 1u), tw & (kFixOne - 1u)) < 5u));
    if (__builtin_expect(!c.fastq || doubtm != 0ull, 0)) {
        // rows 0 and 1 unscaled again (exact: powers of two), then the reference's divisions
        const double unfix = 1.0 / (double)kFixOne;
        const double uu = (ij0 * unfix) / ij2, ww = (ij1 * unfix) / ij2;
        // (int) truncation toward zero + unsigned compare (sdf.cpp:251-256): pixel c is hit by
        // u in (c-1, c+1) for c = 0 and [c, c+1) otherwise; NaN / inf / overflow are rejected.
        const bool inr = uu > -1.0 && uu < (double)c.width && ww > -1.0 && ww < (double)c.height;
        iu1 = inr ? (unsigned)((int)uu + 1) : 1u;
        iw1 = inr ? (unsigned)((int)ww + 1) : 1u;
        inrm = lanes(inr);
    }
    okm &= inrm;
    pixb = __umul24(iu1, c.su) + __umul24(iw1, c.sv);
}

// Correctly rounded n / b for two quotients at a time: hipcc's own sequence for `/` (v_rcp_f32, two Newton FMAs,
// quotient, two residual corrections) WITHOUT its v_div_scale / v_div_fixup wrappers.  The wrappers only act when b,
// 1/b, n/b or a residual leaves the normal range; callers check the operands (div_guard) and use `/` otherwise.
__device__ __forceinline__ v2f rcp_refined(v2f b) {
    v2f r = v2f{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-b, r, v2f{1.0f, 1.0f}), r, r);
}
__device__ __forceinline__ v2f div_core(v2f n, v2f b, v2f r) {
    v2f q = n * r;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(-b, q, n), r, q);
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(-b, q, n), r, q);
    return q;
}
// Range guard of the division core.  Numerators: n * 2^-26 is subnormal exactly for 0 < |n| < 2^-100 (zero stays zero).
// State words (W, Color_W) as integers: 0 <= x < 2^64 <=> bits(x) < bits(2^64) (negative, inf, NaN are larger); the
// new weight added to them lies in [0, 1], so b = state + weight stays within [0, 2^64] -- and b = 0 means n = 0 too
// (both products vanish), which the core turns into the same NaN as 0/0.
__device__ __forceinline__ unsigned long long tiny_lanes(v2f n) {
    const v2f t = n * v2f{0x1p-26f, 0x1p-26f};
    return lanes_subnormal(t.x) | lanes_subnormal(t.y);
}
constexpr unsigned kBits2p64 = 0x5f800000u;

struct GatherState {        // stage 1 done: pixel record requested
    unsigned long long live;   // lane mask (wave-uniform)
    unsigned code;          // the item (wave-uniform)
    double pcx, pcy, pcz;   // camera-frame voxel centre
    u32x4 A, B;             // halves of the pixel records of lanes 0..31 (A) and 32..63 (B): lane l holds half l&1 of the
                            // record of lane l>>1 (A) / 32 + (l>>1) (B)                 (in flight until stage 2)
};
struct UpdateState {        // stage 2 done: volume reads requested
    unsigned long long live;   // lane mask (wave-uniform)
    unsigned code;          // the item (wave-uniform)
    float d_new, w_new;
    unsigned rgb;           // colour: the pixel's packed rgb
    float wc;               // colour: (float)(w_new * cosine), the colour weight
    unsigned off8;          // byte offset of the lane's {D,W} in the item's segment, kDroppedOffset when dead
    u32x2 old;              // {D, W}            (in flight until stage 3)
    u32x4 col;              // {Color_W, R, G, B} (fused colour; in flight until stage 3)
};

#ifndef TSDF_INTEGRATE_SKIP_S2
#define TSDF_INTEGRATE_SKIP_S2 1    // stage 2's un-shuffle and arithmetic only for items with a projected lane (the volume loads stay unconditional)
#endif
#ifndef TSDF_INTEGRATE_COSINE_CORE
#define TSDF_INTEGRATE_COSINE_CORE 1 // the exp() band's f64 cosine by the bare sqrt / division cores when the normal is of ordinary size
#endif
#ifndef TSDF_INTEGRATE_INTERLEAVE
#define TSDF_INTEGRATE_INTERLEAVE 1  // the workgroups of an XCD walk its part of the list together (0: one stretch per workgroup)
#endif
#ifndef TSDF_INTEGRATE_PEEL
#define TSDF_INTEGRATE_PEEL 1       // the software pipeline's fill and drain written out (see the pipeline loop)
#endif
#ifndef TSDF_INTEGRATE_DESC_AHEAD
#define TSDF_INTEGRATE_DESC_AHEAD 1 // the item descriptor of item j+1 is requested at the end of item j's stage 1
#endif
#ifndef TSDF_INTEGRATE_PRIO_LEVEL
#define TSDF_INTEGRATE_PRIO_LEVEL 3
#endif
#ifndef TSDF_INTEGRATE_PRIO_LEVEL_S2
#define TSDF_INTEGRATE_PRIO_LEVEL_S2 1
#endif
#ifndef TSDF_INTEGRATE_PRIO
#define TSDF_INTEGRATE_PRIO 12      // graded wave priorities (s_setprio): bit 2 = the whole of stage 1 at TSDF_INTEGRATE_PRIO_LEVEL, bit 3 = the
#endif                              // whole of stage 2 at ..._LEVEL_S2 (default: 3 / 1, stage 3 at 0); bit 0 = only around stage 1's two gather
                                    // instructions, bit 1 = around the volume loads (measurement builds)
#ifndef TSDF_INTEGRATE_DEPTH
#define TSDF_INTEGRATE_DEPTH 1      // volume reads in flight per wavefront, in items (see the pipeline loop)
#endif
#ifndef TSDF_INTEGRATE_DEBUG
#define TSDF_INTEGRATE_DEBUG 0      // 1 compiles the p.debug timing experiments in
#endif

// weight of sdf.cpp:277-279 for the lanes of the band (others: garbage, dropped by the caller)
template <bool EXPPOLY>
__device__ __forceinline__ float band_weight(float d, float eps) {
    const float a = d - eps;
    const double xarg = (-0.5 * (double)a) * (double)a;
    return EXPPOLY ? (float)exp_taylor8(xarg) : (float)exp(xarg);
}

// floor(n * i / per) for the workgroup's share of its XCD's part of the list (n < 2^26 items, i <= per < 2^12): one f64
// division instead of the 64-bit integer division's ~150 instructions in every wavefront's preamble.  Exact: n * i < 2^38
// is a double, and a quotient that is not an integer lies at least 1 / per away from one -- far outside the division's
// rounding.  Every workgroup uses the same expression, so neighbouring shares meet.
__device__ __forceinline__ unsigned share_split(unsigned n, unsigned i, unsigned per) {
    return (unsigned)(((double)n * (double)i) / (double)per);
}

// The list as list_rows_kernel left it: band b's items in [base[b], base[b] + fill[b]), then the overflow region.
// Segment table (kBins + 1 segments): first VIRTUAL index of each segment (the list without its holes) and what to
// add to a virtual index to get the list entry.  Workgroup 0 also prepares the bookkeeping set of the NEXT launch.
// (first wavefront of the workgroup; the caller's barrier publishes the table)
static_assert(kBins <= 64, "one lane per band in the segment scan");
__device__ __forceinline__ void build_segment_table(const unsigned* __restrict__ set, unsigned* __restrict__ next_set, unsigned ovf_base,
                                                    unsigned long long* __restrict__ totals, unsigned* s_vstart, unsigned* s_delta, int tid,
                                                    unsigned long long* items_out = nullptr /* pinned host: this launch's item count */) {
    if (tid < 64) {
        const bool is_band = tid < kBins;
        const unsigned cur = is_band ? set[kSetCur + tid] : 0u, fo = is_band ? set[kSetFirstOvf + tid] : 0u;
        const unsigned fill = cur < fo ? cur : fo;            // what fitted the band's region
        unsigned incl = fill;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off);
            if (tid >= off) incl += t;
        }
        if (is_band) { s_vstart[tid] = incl - fill; s_delta[tid] = set[kSetBase + tid] - (incl - fill); }
        if (tid == kBins - 1) {
            const unsigned ovf = set[kSetOvf];
            s_vstart[kBins] = incl; s_delta[kBins] = ovf_base - incl;
            s_vstart[kBins + 1] = incl + ovf;
        }
        if (blockIdx.x == 0) {
            // the set of the NEXT launch: capacities from this launch's band counts (+25 % + 128), regions packed from
            // entry 0 and clipped at the overflow region, cursors back to zero
            const unsigned want = is_band ? cur + (cur >> 2) + 128u : 0u;
            unsigned wincl = want;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = __shfl_up(wincl, off);
                if (tid >= off) wincl += t;
            }
            const unsigned wexcl = wincl - want;
            const unsigned base = wexcl < ovf_base ? wexcl : ovf_base;
            const unsigned cap = wexcl < ovf_base ? (want < ovf_base - wexcl ? want : ovf_base - wexcl) : 0u;
            if (is_band) {
                next_set[kSetCur + tid] = 0u; next_set[kSetFirstOvf + tid] = ~0u;
                next_set[kSetCap + tid] = cap; next_set[kSetBase + tid] = base;
            }
            if (tid == kBins - 1) { next_set[kSetBase + kBins] = base + cap; next_set[kSetOvf] = 0u; }
            unsigned long long tot = cur;                    // the launch's item count, for the statistics
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
            if (tid == 0 && tot) atomicAdd(&totals[kCntItems], tot);
            // the host sizes the NEXT launch's grid from it (tsdf_integrate: a wavefront should have >= 16 items)
            if (tid == 0 && items_out) __hip_atomic_store(items_out, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (tid == 0 && set[kSetOvf]) atomicAdd(&totals[kCntOverflowItems], (unsigned long long)set[kSetOvf]);
        }
    }
}

// tsdf_device_frame_released: the launch that packs a frame handed over in device memory tells the host, through a word
// in pinned memory, that the caller's planes have been read -- the first workgroup of the kernel BEHIND the packing
// (integrate_kernel behind list_rows_kernel's appended workgroups; release_kernel behind a pack_kernel launch) stores
// the launch's ticket.  Tickets of one stream grow, the host compares with >=.
__device__ __forceinline__ void publish_release(const ReleaseWord& rel, int tid) {
    if (rel.word && blockIdx.x == 0 && tid == 0)
        __hip_atomic_store(rel.word, rel.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void release_kernel(ReleaseWord rel) { publish_release(rel, (int)threadIdx.x); }
hipError_t launch_release(hipStream_t s, const ReleaseWord& rel) {
    if (!rel.word) return hipSuccess;
    release_kernel<<<dim3(1), dim3(64), 0, s>>>(rel);
    return hipGetLastError();
}

template <bool COLOR, bool KSTD, bool EXPPOLY, bool KTAB>
__global__ __launch_bounds__(kIntegrateBlock, TSDF_INTEGRATE_MIN_WAVES) void integrate_kernel(
    IntegrateParams p, IntegrateTiling tl, const ItemDesc* __restrict__ list, const unsigned* __restrict__ set,
    unsigned* __restrict__ next_set, unsigned ovf_base, unsigned long long* __restrict__ totals,
    float2* __restrict__ dw, float4* __restrict__ crgb, const char* __restrict__ pn,
    unsigned long long* __restrict__ counters /* per workgroup: {owned, halo} updated, cumulative */,
    unsigned* __restrict__ xcd_fb, ReleaseWord rel) {
    constexpr int kRec = COLOR ? 32 : 24, kHalf = kRec / 2;        // bytes of a pixel record / of the piece a lane fetches
    extern __shared__ double s_tab[];
    const int m = p.g.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float delta = p.g.delta, eps = p.g.epsilon, neg_delta = -p.g.delta;
    constexpr unsigned NW = kIntegrateBlock / 64;             // wavefronts per workgroup
    __shared__ unsigned s_vstart[kBins + 2], s_delta[kBins + 1];
    publish_release(rel, tid);      // list_rows_kernel -- and the packing of a device frame in its appended workgroups -- is complete
    build_segment_table(set, next_set, ovf_base, totals, s_vstart, s_delta, tid, rel.items_word);
    if (KTAB) build_k_table(p, s_tab, tid, kIntegrateBlock);
    else __syncthreads();
    const unsigned n_items = __builtin_amdgcn_readfirstlane(s_vstart[kBins + 1]);   // (an LDS load is a per-lane value to the compiler)
    ProjConst pc;
    make_proj_const(p, tl, lane, pc);
    // Workgroups b and b+8 share an XCD (and its 4 MiB L2).  Give each XCD one contiguous eighth of the
    // list = one band of the image, so the pixel records it gathers stay in its own L2.
    // The XCD's part of the list [x_lo, x_hi) follows the shares of update_xcd_shares(); how its gridDim.x / 8 workgroups
    // share it: below.
    const unsigned xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    // (readfirstlane: xcd_fb is written at the end of this kernel, so the compiler may fetch the shares with a vector
    // load and then takes everything derived from them for per-lane values.  Requesting them in front of the segment
    // table's loads changes nothing: measured.)
    const unsigned x_lo = (unsigned)(((unsigned long long)n_items * (unsigned)__builtin_amdgcn_readfirstlane((int)xcd_fb[xcd])) >> 24);
    const unsigned x_hi = (unsigned)(((unsigned long long)n_items * (unsigned)__builtin_amdgcn_readfirstlane((int)xcd_fb[xcd + 1])) >> 24);
    // Items are dealt to the wavefronts of a workgroup ITEM BY ITEM: at any moment its NW wavefronts work on NW consecutive
    // items, i.e. on neighbouring voxel rows.
#if TSDF_INTEGRATE_INTERLEAVE
    // The XCD's workgroups walk its part of the list TOGETHER: workgroup w takes the items [x_lo + (j * per_xcd + w) * NW,
    // + NW) for j = 0, 1, ...  Neighbouring items cost alike (12 % of the items update nothing, a quarter touches 1-16
    // voxels, and they come in runs), so with one contiguous stretch per workgroup (rounds 1-3) stretches of equal length
    // differed in cost: the wavefronts' item loops ended 101 us into the launch at the latest and after 82 on average
    // (tools/wg_finish_probe.py).  Every workgroup now gets an even sample of the band -- 95 us at the latest -- and the
    // XCD's 160 workgroups touch the same part of the image at the same time.  -2 % at 512^3, -2.5 % at 1024^3, both
    // scenes, with and without colour (r04_integrate_fixed_costs.json; what did NOT work there: item pools with atomic
    // cursors, shares per workgroup generation, shares per workgroup by feedback).
    const unsigned v_stride = per_xcd * NW, v_first = x_lo + in_xcd * NW + (unsigned)wv, v_lim = x_hi;
    const int cnt = __builtin_amdgcn_readfirstlane(v_first < x_hi ? (int)share_split(x_hi - v_first + v_stride - 1u, 1u, v_stride) : 0);
#else
    // one contiguous stretch of the XCD's part per workgroup
    const unsigned wg_first = (unsigned)__builtin_amdgcn_readfirstlane((int)(x_lo + share_split(x_hi - x_lo, in_xcd, per_xcd)));
    const unsigned wg_last = (unsigned)__builtin_amdgcn_readfirstlane((int)(x_lo + share_split(x_hi - x_lo, in_xcd + 1u, per_xcd)));
    const unsigned v_stride = NW, v_first = wg_first + (unsigned)wv, v_lim = wg_last;
    const int cnt = wg_last > v_first ? (int)((wg_last - v_first + NW - 1u) / NW) : 0;
#endif
    unsigned n_own = 0, n_halo = 0;
    const unsigned long long loop_t0 = __builtin_amdgcn_s_memrealtime();      // for the XCD shares of the next launch

    // planes of the frame's pixel data, their bases shifted by the bias of the record index
    const long long npix = (long long)p.width * p.height;
    const long long bias = (long long)pc.su + pc.sv;
    const __amdgpu_buffer_rsrc_t pn_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(pn - bias * kRec), 0, (int)((npix + bias) * kRec), kRsrcWord3);
    const unsigned lane8 = (unsigned)lane * 8u;
    const unsigned half_off = (unsigned)(lane & 1) * (unsigned)kHalf;
    __shared__ u32x4 s_pieces[kIntegrateBlock / 64][128];      // wave-private un-shuffle buffer of the paired gather
    const unsigned dropped = kDroppedOffset;
    const unsigned own_row0 = (unsigned)((p.g.own_x0 - p.g.xs) * m), own_row1 = (unsigned)((p.g.own_x1 - p.g.xs) * m);   // rows < 2^26

    // One pipeline step = S1(j) | S2(j-1) | S3(j-1-DEPTH): three independent instruction streams.
    // S1 requests the pixel record of item j, S2 consumes the record requested a step earlier and requests the volume data,
    // S3 consumes the volume data requested DEPTH steps earlier -- every request has (at least) a whole step to complete.
    // virtual index -> list entry: the wavefront walks its items in increasing order and keeps the segment it is in.
    // (seg_end is wave-uniform in an SGPR; seg_delta stays in a VGPR as the LDS load leaves it -- the kernel has no
    // scalar registers to spare -- and goes through v_readfirstlane once per item.)
    unsigned seg_end = 0u, seg_delta = 0u;
#ifdef TSDF_LIVE_HISTOGRAM
    unsigned hist[6] = {0u, 0u, 0u, 0u, 0u, 0u};              // measurement build: items by updated lanes (0, 1-16, 17-32, 33-48, 49-63, 64)
#endif
    auto locate = [&](unsigned v) {
        // segment of v = number of segment ends <= v (lane l looks at the end of segment l; the overflow segment is the last)
        const unsigned e = s_vstart[(lane < kBins ? lane : kBins - 1) + 1];
        const unsigned sg = (unsigned)__popcll(__ballot(lane < kBins && v >= e));
        seg_end = __builtin_amdgcn_readfirstlane(s_vstart[sg + 1]);
        seg_delta = s_delta[sg];
    };
    // One item's descriptor: wave-uniform, one scalar 32-byte load (no item left: entry 0, masked by the caller).
    auto fetch_desc = [&](unsigned v) {
        unsigned entry = 0u;
        if (v < v_lim) {
            if (__builtin_expect(v >= seg_end, 0)) locate(v);
            entry = v + seg_delta;
        }
        return list[__builtin_amdgcn_readfirstlane(entry)];
    };
    auto item_v = [&](int j) { return v_first + v_stride * (unsigned)j; };           // virtual index of this wavefront's item j
#if TSDF_INTEGRATE_DESC_AHEAD
    // ... requested one item ahead: stage 1 used to open with the load and an s_waitcnt lgkmcnt(0) right behind it -- a
    // trip to L2 at the stage's raised priority in front of every item
    ItemDesc dnext = fetch_desc(item_v(0));
#endif
    auto stage1 = [&](int j, GatherState& g /*out: item j*/) {
        // Stage 1 runs at raised wave priority: it ends in the gathers, the longest trip of an item (64 scattered records
        // through L1 / L2), and a wavefront on its way to them should not queue behind the arithmetic of its four
        // neighbours on the SIMD.  Measured on five boxes, alternating builds: with the priority only around the two gather
        // instructions integrate_kernel is 2-7 % shorter on four of them (113.5 -> 105.7-110.4 us, 113.7 -> 110.2-111.0,
        // 113.4 -> 110.4, 108.9 -> 106.4) and sits on two levels (106.3 / 110.5 against 108.7) on the fifth; the whole
        // stage takes another 1.0-1.2 us (109.1-109.5); stage 2 at the SAME priority makes it worse, stage 2 one step above
        // stage 3 (3 / 1 / 0) another ~3 us (109.3 -> 106.4 in 5 of 5 alternations on one box; 105.0 -> 101.2-102.1 in two
        // of three on another, 106.1 in the third).
#if TSDF_INTEGRATE_PRIO & 4
        __builtin_amdgcn_s_setprio(TSDF_INTEGRATE_PRIO_LEVEL);
#endif
        const bool have = j < cnt;
#if TSDF_INTEGRATE_DESC_AHEAD
        const ItemDesc ds = dnext;                              // requested at the end of the previous item's stage 1
#else
        const ItemDesc ds = fetch_desc(item_v(j));
#endif
        unsigned long long okm;
        unsigned pixb;
        project_item<KSTD, KTAB>(pc, ds, s_tab, lane, g.pcx, g.pcy, g.pcz, okm, pixb);
        if (!have) okm = 0ull;
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 1) pixb = __builtin_amdgcn_readfirstlane(pixb);       // timing experiment only: one record per wave
#endif
        // Pixel-record gather, paired: the vector L1 serves a wave's gather about one lane-address at a time, and the
        // two halves of a record are two instructions.  Instead the first load fetches both halves of
        // the records of lanes 0..31 (lane l: record of lane l/2, half l%2), the second those of lanes 32..63: lane pairs
        // share a line, so the look-ups of an item are halved.  Stage 2 un-shuffles the pieces through a wave-private
        // LDS buffer.  Dead lanes carry an offset beyond the plane: no look-up at all.
        const unsigned roff = select_by_mask(okm, COLOR ? pixb << 5 : __umul24(pixb, (unsigned)kRec), dropped);
        const unsigned ra = (unsigned)__shfl((int)roff, lane >> 1) + half_off;
        const unsigned rb = (unsigned)__shfl((int)roff, 32 + (lane >> 1)) + half_off;
#if TSDF_INTEGRATE_PRIO & 1
        __builtin_amdgcn_s_setprio(TSDF_INTEGRATE_PRIO_LEVEL);
#endif
        if (COLOR) {
            g.A = __builtin_amdgcn_raw_buffer_load_b128(pn_rsrc, (int)ra, 0, 0);      // piece for LDS slot lane
            g.B = __builtin_amdgcn_raw_buffer_load_b128(pn_rsrc, (int)rb, 0, 0);      // piece for LDS slot 64 + lane
        } else {
            const u32x3 a3 = __builtin_amdgcn_raw_buffer_load_b96(pn_rsrc, (int)ra, 0, 0);
            const u32x3 b3 = __builtin_amdgcn_raw_buffer_load_b96(pn_rsrc, (int)rb, 0, 0);
            g.A = u32x4{a3.x, a3.y, a3.z, 0u}; g.B = u32x4{b3.x, b3.y, b3.z, 0u};
        }
#if TSDF_INTEGRATE_DESC_AHEAD
        dnext = fetch_desc(item_v(j + 1));
#endif
#if TSDF_INTEGRATE_PRIO & 5
        __builtin_amdgcn_s_setprio(0);
#endif
        g.live = okm;
        g.code = ds.code;
    };
    auto stage2 = [&](const GatherState& gin /*item j-1, record arrived*/, UpdateState& u /*out: item j-1*/) {
#if TSDF_INTEGRATE_PRIO & 8
        __builtin_amdgcn_s_setprio(TSDF_INTEGRATE_PRIO_LEVEL_S2);
#endif
        float d = 0.f, wn = 1.0f, wc = 0.f;
        unsigned rgbv = 0u;
        unsigned long long okm = 0ull;
        // An item none of whose lanes projects into the image skips the un-shuffle and the distances; like stage 3's skip
        // this leaves the vector-memory operations of a step where they are.  Putting the volume loads or the stores of
        // items that update nothing behind the same kind of branch costs 3-10 us: hipcc then waits for the smallest
        // outstanding count at every use (r04_integrate_fixed_costs.json).
#if TSDF_INTEGRATE_SKIP_S2
        if (gin.live != 0ull)
#endif
        {
        u32x4* stage = s_pieces[wv];
        if (COLOR) { stage[lane] = gin.A; stage[64 + lane] = gin.B; }
        else {
            *reinterpret_cast<u32x3*>(&stage[lane]) = u32x3{gin.A.x, gin.A.y, gin.A.z};
            *reinterpret_cast<u32x3*>(&stage[64 + lane]) = u32x3{gin.B.x, gin.B.y, gin.B.z};
        }
        // other LANES read what this lane wrote: the compiler's memory model is per thread, so without a
        // wavefront-scope fence it may (and did) hoist the reads above the second write.  No instruction is
        // emitted: LDS operations of one wave execute in order.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x4 P, N;                                            // own record: {Px,Py,Pz,rgb} {Nx,Ny,Nz,(float)cosine}
        if (COLOR) { P = stage[2 * lane + 0]; N = stage[2 * lane + 1]; }
        else {
            const u32x3 p3 = *reinterpret_cast<const u32x3*>(&stage[2 * lane + 0]), n3 = *reinterpret_cast<const u32x3*>(&stage[2 * lane + 1]);
            P = u32x4{p3.x, p3.y, p3.z, 0u}; N = u32x4{n3.x, n3.y, n3.z, 0u};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads above stay before the next step's writes
        __builtin_amdgcn_wave_barrier();
        const float Px = __uint_as_float(P.x), Py = __uint_as_float(P.y), Pz = __uint_as_float(P.z);
        const float Nx = __uint_as_float(N.x), Ny = __uint_as_float(N.y), Nz = __uint_as_float(N.z);
        // sdf.cpp:260: NaN in P.x, P.y or the normal
        const unsigned long long nanm = lanes(__builtin_isunordered(Px, Py)) | lanes(__builtin_isunordered(Nx, Ny)) | lanes(is_nan(Nz));
        // projectivePointToPlaneDistance, sdf.h:177-181 (Eigen dot: a0*b0 + (a1*b1 + a2*b2))
        const double dx = (double)Px - gin.pcx, dy = (double)Py - gin.pcy, dz = (double)Pz - gin.pcz;
        const double p2p = dx * (double)Nx + (dy * (double)Ny + dz * (double)Nz);
        d = (float)p2p;                                               // sdf.cpp:274
        okm = gin.live & ~nanm & ~lanes(d > delta);      // sdf.cpp:280-283
        const unsigned long long bandm = okm & lanes(d >= eps);             // sdf.cpp:277-279 (d <= delta holds in okm)
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 2) okm = 0ull;                                        // timing experiment only: no volume RMW
#endif
        // sdf.cpp:294-299: wc = (float)(w_new * cosine).  For w_new == 1 that is the pre-rounded cosine of the record;
        // a wavefront with lanes in the exp() band recomputes the f64 cosine from the normal for those lanes.
        wc = COLOR ? __uint_as_float(N.w) : 0.f;
        if (bandm != 0ull) {
            wn = __uint_as_float(select_by_mask(bandm, __float_as_uint(band_weight<EXPPOLY>(d, eps)), 0x3f800000u));
            if (COLOR) {
#if TSDF_INTEGRATE_COSINE_CORE
                const double nxd = (double)Nx, nyd = (double)Ny, nzd = (double)Nz;
                const double n2 = nxd * nxd + (nyd * nyd + nzd * nzd);
                const unsigned long long plain = lanes(n2 >= 0x1p-200) & lanes(n2 <= 0x1p200);
                double cosine;
                if (__builtin_expect((bandm & ~plain) == 0ull, 1)) cosine = pixel_cosine_core(nzd, n2);
                else cosine = pixel_cosine(Nx, Ny, Nz);
#else
                const double cosine = pixel_cosine(Nx, Ny, Nz);
#endif
                wc = __uint_as_float(select_by_mask(bandm, __float_as_uint((float)((double)wn * cosine)), __float_as_uint(wc)));
            }
        }
        rgbv = P.w;
        }
        if (COLOR) { u.rgb = rgbv; u.wc = wc; }
        d = d < neg_delta ? neg_delta : d;                                  // sdf.cpp:285-287
        u.d_new = d; u.w_new = wn;
        u.live = okm;
        const unsigned code2 = gin.code;
        u.code = code2;
        long long base2 = (long long)(code2 >> 6) * m + (long long)(code2 & 63u) * 64;
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 64) base2 = 0;                                        // timing experiment only: cache-resident volume reads
#endif
        u.off8 = select_by_mask(okm, lane8, dropped);
        const __amdgpu_buffer_rsrc_t seg_dw = __builtin_amdgcn_make_buffer_rsrc(dw + base2, 0, 64 * (int)sizeof(float2), kRsrcWord3);
        const unsigned ld8 = u.off8;
        u.old = __builtin_amdgcn_raw_buffer_load_b64(seg_dw, (int)ld8, 0, 0);   // {D,W}: the tracker re-reads these lines -> keep them cached
        if (COLOR) {   // colour is streamed once per frame and never read by the tracker: non-temporal
            const __amdgpu_buffer_rsrc_t seg_c = __builtin_amdgcn_make_buffer_rsrc(crgb + base2, 0, 64 * (int)sizeof(float4), kRsrcWord3);
            u.col = __builtin_amdgcn_raw_buffer_load_b128(seg_c, (int)(ld8 << 1), 0, 2);
        }
#if TSDF_INTEGRATE_PRIO & 8
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto stage3 = [&](const UpdateState& uin /*item j-1-DEPTH, volume data arrived*/) {
        const unsigned code3 = uin.code;
        const unsigned row3 = code3 >> 6;
        const bool owned3 = row3 >= own_row0 && row3 < own_row1;            // wave-uniform; rows of the owned x layers
        long long base3 = (long long)row3 * m + (long long)(code3 & 63u) * 64;
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 1024) base3 = (long long)(blockIdx.x & 255) * 4096 + wv * 64;   // timing experiment only: stores to a cache-resident region
#endif
        const unsigned n_live = (unsigned)__popcll(uin.live);
#ifdef TSDF_LIVE_HISTOGRAM
        if (uin.code != 0u || n_live) { const unsigned bk = n_live == 0u ? 0u : n_live == 64u ? 5u : 1u + ((n_live - 1u) >> 4); hist[bk] += 1u; }
#endif
        n_own += owned3 ? n_live : 0u;
        n_halo += owned3 ? 0u : n_live;
        // sdf.cpp:289-292 (D, W) and :294-304 (colour), as packed f32 pairs {D-average, R} and {G, B}.  One item in eight
        // updates nothing (r04_integrate_fixed_costs.json): its averages are skipped -- the stores below stay where they are
        // (every lane's offset is out of range), so the count of vector-memory operations per step does not change.
        v2f sum1 = v2f{0.f, 0.f}, q1 = v2f{0.f, 0.f}, q2 = v2f{0.f, 0.f};
        if (uin.live != 0ull) {
            const float W = __uint_as_float(uin.old.y), D = __uint_as_float(uin.old.x);
            const float cx = __uint_as_float(uin.col.x);
            const float wc = COLOR ? uin.wc : 0.f;
            v2f num1, num2 = v2f{0.f, 0.f};
            sum1.x = W + uin.w_new;
            num1.x = W * D + uin.w_new * uin.d_new;
            if (COLOR) {
                const unsigned rgb = uin.rgb;
                const float pr = (float)(rgb & 255u), pg = (float)((rgb >> 8) & 255u), pb = (float)((rgb >> 16) & 255u);
                sum1.y = cx + wc;
                num1.y = cx * __uint_as_float(uin.col.y) + wc * pr;
                num2 = v2f{cx, cx} * v2f{__uint_as_float(uin.col.z), __uint_as_float(uin.col.w)} + v2f{wc, wc} * v2f{pg, pb};
            } else {
                sum1.y = 1.0f; num1.y = 0.0f;
            }
            const v2f r = rcp_refined(sum1);
            q1 = div_core(num1, sum1, r);
            if (COLOR) q2 = div_core(num2, v2f{sum1.y, sum1.y}, v2f{r.y, r.y});
            unsigned long long bad = tiny_lanes(num1) | lanes(uin.old.y >= kBits2p64);
            if (COLOR) bad |= tiny_lanes(num2) | lanes(uin.col.x >= kBits2p64);
            if (__builtin_expect((bad & uin.live) != 0ull, 0)) {
                q1.x = num1.x / sum1.x;
                if (COLOR) { q1.y = num1.y / sum1.y; q2.x = num2.x / sum1.y; q2.y = num2.y / sum1.y; }
            }
        }
        const __amdgpu_buffer_rsrc_t seg_dw = __builtin_amdgcn_make_buffer_rsrc(dw + base3, 0, 64 * (int)sizeof(float2), kRsrcWord3);
        u32x2 o2; o2.x = __float_as_uint(q1.x); o2.y = __float_as_uint(sum1.x);
        unsigned off8 = uin.off8;
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 32) off8 = dropped;                                   // timing experiment only: no stores
#endif
        __builtin_amdgcn_raw_buffer_store_b64(o2, seg_dw, (int)off8, 0, 0);
        if (COLOR) {
            const __amdgpu_buffer_rsrc_t seg_c = __builtin_amdgcn_make_buffer_rsrc(crgb + base3, 0, 64 * (int)sizeof(float4), kRsrcWord3);
            u32x4 c4; c4.x = __float_as_uint(sum1.y); c4.y = __float_as_uint(q1.y); c4.z = __float_as_uint(q2.x); c4.w = __float_as_uint(q2.y);
            __builtin_amdgcn_raw_buffer_store_b128(c4, seg_c, (int)(off8 << 1), 0, 2);   // dropped stays out of range; nt: colour is streamed
        }

    };

    // Software pipeline over the wavefront's items, unrolled over one full rotation of the state registers so
    // that in-flight registers never have to be copied (a copy would force the wait):
    //   step j:  S1(j) request pixel record | S2(j-1) request {D,W}/colour | S3(j-1-DEPTH) average + store
    // DEPTH = steps between the volume request of an item and its use: DEPTH + 1 update states rotate, i.e.
    // DEPTH items' worth of HBM reads stay in flight per wavefront.
    constexpr int NU = TSDF_INTEGRATE_DEPTH + 1, NG = 2;
    constexpr int PERIOD = (NU % 2 == 0) ? NU : 2 * NU;
    GatherState G[NG];
    UpdateState U[NU];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        G[q].live = 0ull; G[q].code = 0u;
        G[q].pcx = G[q].pcy = G[q].pcz = 0.0;
        G[q].A = u32x4{0u, 0u, 0u, 0u}; G[q].B = G[q].A;
    }
#pragma unroll
    for (int q = 0; q < NU; ++q) {
        U[q].live = 0ull; U[q].code = 0u; U[q].rgb = 0u; U[q].wc = 0.f;
        U[q].d_new = 0.f; U[q].w_new = 1.f; U[q].off8 = kDroppedOffset;
        U[q].old = u32x2{0u, 0x3f800000u}; U[q].col = u32x4{0x3f800000u, 0u, 0u, 0u};
    }
#if TSDF_INTEGRATE_DEBUG
    // stage profile (bit 256): the first wavefront of every workgroup stamps the shader clock at the stage boundaries
    // (s_memtime + s_waitcnt lgkmcnt(0), a scheduling barrier for memory operations) and adds up what each stage took
    unsigned long long t_s1 = 0, t_s2 = 0, t_s3 = 0, n_steps = 0;
    const bool stamp = (p.debug & 256) != 0 && ((p.debug & 512) != 0 || wv == 0);      // bit 512: every wavefront stamps
    auto clock_now = [&]() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); return t; };
    unsigned long long rt0 = 0, ct0 = 0;
    if (stamp) { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) : : "memory"); ct0 = clock_now(); }
#endif
#if TSDF_INTEGRATE_PEEL && !TSDF_INTEGRATE_DEBUG
    // The pipeline's fill and drain, peeled: the rolled loop below runs cnt + 2 (or 3) steps of three stages each, i.e.
    // 2-3 steps' worth of stages on items that do not exist -- with ~38 items per wavefront that is 6 % of all issued
    // instructions.  While the kernel's stages queued behind each other's arithmetic that changed nothing (measured in
    // the round's first half); with the graded priorities the kernel sits within ~10 % of its issue floor and the peeled
    // form is worth 0.8 us (105.3 -> 104.5 us, 4 of 4 alternations).  The steady part is the same two-step rotation.
    static_assert(TSDF_INTEGRATE_DEPTH == 1, "the peeled pipeline is written for two rotating update states");
    if (cnt >= 2) {
        stage1(0, G[0]);
        stage1(1, G[1]); stage2(G[0], U[1]);
        int j = 2;
        for (; j + 1 < cnt; j += 2) {
            stage1(j, G[0]);     stage2(G[1], U[0]); stage3(U[1]);
            stage1(j + 1, G[1]); stage2(G[0], U[1]); stage3(U[0]);
        }
        if (j < cnt) {           // an odd item count: one more full step, then the drain
            stage1(j, G[0]); stage2(G[1], U[0]); stage3(U[1]);
            stage2(G[0], U[1]); stage3(U[0]);
            stage3(U[1]);
        } else {
            stage2(G[1], U[0]); stage3(U[1]);
            stage3(U[0]);
        }
    } else
#endif
    for (int j = 0; j < cnt + 1 + TSDF_INTEGRATE_DEPTH; j += PERIOD) {
#pragma unroll
        for (int q = 0; q < PERIOD; ++q) {
            // S1(j+q) -> G[q%2];  S2(j+q-1): G[(q+1)%2] -> U[q%NU];  S3(j+q-1-DEPTH): U[(q+1)%NU] (the oldest)
#if TSDF_INTEGRATE_DEBUG
            if (stamp) {
                const unsigned long long c0 = clock_now();
                stage1(j + q, G[q % NG]);
                const unsigned long long c1 = clock_now();
                stage2(G[(q + 1) % NG], U[q % NU]);
                const unsigned long long c2 = clock_now();
                stage3(U[(q + 1) % NU]);
                const unsigned long long c3 = clock_now();
                t_s1 += c1 - c0; t_s2 += c2 - c1; t_s3 += c3 - c2; ++n_steps;
                continue;
            }
#endif
            stage1(j + q, G[q % NG]);
            stage2(G[(q + 1) % NG], U[q % NU]);
            stage3(U[(q + 1) % NU]);
        }
    }
#if TSDF_INTEGRATE_DEBUG
    if (stamp && lane == 0) {
        unsigned long long rt1, ct1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) : : "memory");
        ct1 = clock_now();
        // per workgroup and wavefront: cycles in S1, S2, S3, steps, loop cycles, loop time in 10 ns ticks
        unsigned long long* w = counters + 2 * (size_t)gridDim.x + 6 * ((size_t)blockIdx.x * NW + (size_t)wv);
        w[0] += t_s1; w[1] += t_s2; w[2] += t_s3; w[3] += n_steps; w[4] += ct1 - ct0; w[5] += rt1 - rt0;
    }
#endif
    // (steps run up to j >= cnt + DEPTH, so S3 has retired item cnt-1 inside the loop: nothing to drain)
#ifdef TSDF_LIVE_HISTOGRAM
    if (lane == 0) {
        unsigned long long* w = counters + 2 * (size_t)gridDim.x + 6 * ((size_t)blockIdx.x * NW + (size_t)wv);
        for (int q = 0; q < 6; ++q) w[q] += hist[q];
    }
#endif

#ifdef TSDF_WG_FINISH
    if (lane == 0) {     // measurement build: when this wavefront's item loop began and ended (100 MHz ticks), last launch
        unsigned long long* w = counters + 2 * (size_t)gridDim.x + 6 * ((size_t)blockIdx.x * NW + (size_t)wv);
        w[0] = loop_t0; w[1] = __builtin_amdgcn_s_memrealtime(); w[2] = (unsigned long long)cnt;
    }
#endif
    if (wv == 0 && lane == 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(xcd_fb + kFbTicksWord) + xcd, __builtin_amdgcn_s_memrealtime() - loop_t0 + 1ull);
    // Update counts (wave-uniform already): LDS across the waves, then the workgroup adds to ITS OWN pair of cumulative
    // words (plain read-modify-write, nobody else touches them; the host adds the pairs up when somebody asks).
    __shared__ unsigned s_cnt[2][kIntegrateBlock / 64];
    if (lane == 0) { s_cnt[0][tid >> 6] = n_own; s_cnt[1][tid >> 6] = n_halo; }
    __syncthreads();
    if (tid == 0) {
        unsigned a = 0, b = 0;
        for (int q = 0; q < kIntegrateBlock / 64; ++q) { a += s_cnt[0][q]; b += s_cnt[1][q]; }
        if (a) counters[2 * blockIdx.x + 0] += (unsigned long long)a;
        if (b) counters[2 * blockIdx.x + 1] += (unsigned long long)b;
    }
}

// ---- integrate_queue_kernel (round 4) -------------------------------------------------------------------------------
//
// integrate_kernel pays every instruction of an item for 64 lanes of which about 33 are updated (plant scene, 512^3:
// 199.9 k items, 12.8 M listed voxels, 10.2 M inside the frustum, 6.33 M updated; 45 % of the items have voxels in the
// exp() band and run its f64 weight + cosine for a handful of lanes).  Its cost follows the number of items, not the
// bytes (+36 % voxels -> +12 % time at the same item count; 6.6 vector-memory instructions per item).
// Here the part of the work that only updated voxels need runs on DENSE wavefronts:
//   per item (64 lanes)   S1 geometry + pixel-record gather, S2 distance and the tests of sdf.cpp:260-283; the lanes
//                         that pass append {voxel index, d, colour weight | pixel, rgb} (16 bytes; 8 without colour) to
//                         one of two wave-private queues in LDS (ballot + mbcnt prefix): weight-1 lanes (d < epsilon)
//                         and lanes of the exp() band;
//   per 64 queued lanes   a BATCH: pop 64 entries, (band queue: exp() weight, and the f64 cosine of the pixel from the
//                         plane pack_kernel wrote -- one 8-byte gather per 64 band voxels), {D,W} + colour loads,
//                         the running averages of sdf.cpp:289-304 and the stores, all 64 lanes live.
// Every voxel still belongs to exactly one item, each updated voxel is queued exactly once and updated with the
// reference's operations in the reference's order, so the volume is bit-identical to integrate_kernel's; only the
// grouping of voxels into wavefront instructions differs.  Wave-private queues: no barrier, no atomics; LDS operations
// of one wavefront execute in order.  A queue never holds more than 63 + 64 entries: after an item at most one queue can
// have reached 64 unless both were nearly full, and the consume loop pops until both are below 64 again.
// Volume accesses are raw buffer operations with 32-bit offsets into a WINDOW of 2^27 voxels (1 GiB of {D,W}, 2 GiB of
// colour; a 512^3 volume is one window): the queues hold window-relative voxel indices, an item of another window than
// the current one first flushes them (a wavefront's items follow the row order of the list, so this is rare), and lanes
// without a voxel carry an out-of-range offset and touch no memory -- which keeps every vector-memory operation of the
// batch stages unconditional (see form_batch).

#ifndef TSDF_QUEUE_ALIGN
#define TSDF_QUEUE_ALIGN 1                // lanes are queued in aligned groups of this many (1: lane by lane)
#endif
#ifndef TSDF_QUEUE_FIFO
#define TSDF_QUEUE_FIFO 1                 // queues are rings (voxels leave in the order they came: a batch = the lanes of neighbouring items)
#endif
constexpr int kQCap = 128;                 // entries per queue and wavefront (>= 63 + 64)
constexpr int kWindowBits = 27;            // voxels per volume window: offsets (index << 4 for colour) stay below 2^31
constexpr unsigned kDroppedVolumeOffset = 0xfffffff0u;   // beyond every window, also when doubled for the colour array
[[maybe_unused]] constexpr unsigned kHoleVoxel = 0x1ffffffeu;              // queue entry without a voxel (aligned compaction): (x << 3) is out of range

template <bool COLOR> struct QueueEntry { typedef u32x4 T; };     // {voxel, d bits, (float)cosine | biased pixel index, rgb}
template <> struct QueueEntry<false> { typedef u32x2 T; };        // {voxel, d bits}

template <bool COLOR, bool KSTD, bool EXPPOLY, bool KTAB>
__global__ __launch_bounds__(kIntegrateBlock, 4) void integrate_queue_kernel(
    IntegrateParams p, IntegrateTiling tl, const ItemDesc* __restrict__ list, const unsigned* __restrict__ set,
    unsigned* __restrict__ next_set, unsigned ovf_base, unsigned long long* __restrict__ totals,
    float2* __restrict__ dw, float4* __restrict__ crgb, const char* __restrict__ pn,
    unsigned long long* __restrict__ counters /* per workgroup: {owned, halo} updated, cumulative */,
    unsigned* __restrict__ xcd_fb, ReleaseWord rel) {
    typedef typename QueueEntry<COLOR>::T Entry;
    publish_release(rel, (int)threadIdx.x);
    constexpr int kRec = COLOR ? 32 : 24, kHalf = kRec / 2;        // bytes of a pixel record / of the piece a lane fetches
    extern __shared__ double s_tab[];
    const int m = p.g.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float delta = p.g.delta, eps = p.g.epsilon, neg_delta = -p.g.delta;
    constexpr unsigned NW = kIntegrateBlock / 64;             // wavefronts per workgroup
    __shared__ unsigned s_vstart[kBins + 2], s_delta[kBins + 1];
    build_segment_table(set, next_set, ovf_base, totals, s_vstart, s_delta, tid, rel.items_word);
    if (KTAB) build_k_table(p, s_tab, tid, kIntegrateBlock);
    else __syncthreads();
    const unsigned n_items = __builtin_amdgcn_readfirstlane(s_vstart[kBins + 1]);
    ProjConst pc;
    make_proj_const(p, tl, lane, pc);
    // the XCD's part of the list and the workgroup's share of it: as integrate_kernel
    const unsigned xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const unsigned x_lo = (unsigned)(((unsigned long long)n_items * (unsigned)__builtin_amdgcn_readfirstlane((int)xcd_fb[xcd])) >> 24);
    const unsigned x_hi = (unsigned)(((unsigned long long)n_items * (unsigned)__builtin_amdgcn_readfirstlane((int)xcd_fb[xcd + 1])) >> 24);
    const unsigned wg_first = (unsigned)__builtin_amdgcn_readfirstlane((int)(x_lo + share_split(x_hi - x_lo, in_xcd, per_xcd)));
    const unsigned wg_last = (unsigned)__builtin_amdgcn_readfirstlane((int)(x_lo + share_split(x_hi - x_lo, in_xcd + 1u, per_xcd)));
    const int cnt = wg_last > wg_first + (unsigned)wv ? (int)((wg_last - wg_first - (unsigned)wv + NW - 1u) / NW) : 0;
    unsigned n_own = 0, n_halo = 0;
    const unsigned long long loop_t0 = __builtin_amdgcn_s_memrealtime();      // for the XCD shares of the next launch

    const long long npix = (long long)p.width * p.height;
    const long long bias = (long long)pc.su + pc.sv;
    const __amdgpu_buffer_rsrc_t pn_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(pn - bias * kRec), 0, (int)((npix + bias) * kRec), kRsrcWord3);
    // the f64 cosines behind the records (colour volumes), addressed by the same biased record index
    const __amdgpu_buffer_rsrc_t cos_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(pn + npix * (long long)kPixelRecordBytes - bias * 8), 0, (int)((npix + bias) * 8), kRsrcWord3);
    const unsigned half_off = (unsigned)(lane & 1) * (unsigned)kHalf;
    __shared__ u32x4 s_pieces[NW][128];                      // wave-private un-shuffle buffer of the paired gather
    __shared__ Entry s_queue[NW][2][kQCap];                  // wave-private queues: [0] weight-1 lanes, [1] exp()-band lanes
    Entry* const q_plain = s_queue[wv][0];
    Entry* const q_band = s_queue[wv][1];
    const unsigned dropped = kDroppedOffset;
    const unsigned own_row0 = (unsigned)((p.g.own_x0 - p.g.xs) * m), own_row1 = (unsigned)((p.g.own_x1 - p.g.xs) * m);   // rows < 2^26

    unsigned seg_end = 0u, seg_delta = 0u;
    auto locate = [&](unsigned v) {
        const unsigned e = s_vstart[(lane < kBins ? lane : kBins - 1) + 1];
        const unsigned sg = (unsigned)__popcll(__ballot(lane < kBins && v >= e));
        seg_end = __builtin_amdgcn_readfirstlane(s_vstart[sg + 1]);
        seg_delta = s_delta[sg];
    };

    struct Gather {             // S1 done: pixel record requested
        unsigned long long live;   // lanes that pass the geometric tests (wave-uniform mask)
        unsigned code;          // the item (wave-uniform)
        double pcx, pcy, pcz;   // camera-frame voxel centre
        unsigned pixb;          // biased record index of the lane's pixel
        u32x4 A, B;             // halves of the pixel records, as in integrate_kernel (in flight until S2)
    };
    struct Batch {              // 64 queued voxels whose volume data is in flight
        unsigned long long valid;  // lanes that hold a voxel (all of them or none, except when the queues are drained at the end)
        bool band;              // wave-uniform: the voxels come from the exp()-band queue (their weights are formed when the batch is finished)
        unsigned off8;          // byte offset of the voxel's {D,W} in the current window (lanes without a voxel: out of range)
        float d_new, wc;        // wc: the colour weight of a weight-1 voxel
        unsigned rgb;
        u32x2 cosb;             // band voxels: the f64 cosine of the pixel (in flight)
        u32x2 old;              // {D, W}             (in flight)
        u32x4 col;              // {Color_W, R, G, B} (in flight)
    };
    // wave-uniform queue state: `fill` entries are queued (FIFO: a ring starting at `head`; otherwise a stack)
    unsigned fill_p = 0u, fill_b = 0u, head_p = 0u, head_b = 0u;
    // Two batches rotate: the volume data of the batch formed in step j is requested BEFORE the batch of step j-1 is
    // finished and stored, so that no wait for loads ever has to sit behind a store (a wavefront's vector-memory
    // operations complete in order: with the stores in front of the next batch's loads every batch waited for the
    // previous one's stores to be acknowledged -- 50 us of the launch).
    Batch BB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        BB[q].valid = 0ull; BB[q].band = false; BB[q].off8 = kDroppedVolumeOffset; BB[q].d_new = 0.f; BB[q].wc = 0.f; BB[q].rgb = 0u;
        BB[q].cosb = u32x2{0u, 0u}; BB[q].old = u32x2{0u, 0x3f800000u}; BB[q].col = u32x4{0x3f800000u, 0u, 0u, 0u};
    }
    // the window of the volume the queued voxels belong to, and its two buffer resources
    const unsigned long long n_stored = (unsigned long long)(p.g.xe - p.g.xs) * (unsigned)m * (unsigned)m;
    const bool multi_window = n_stored > (1ull << kWindowBits);
    unsigned w_cur = 0u;
    __amdgpu_buffer_rsrc_t dw_rsrc, col_rsrc;
    auto set_window = [&](unsigned w) {
        const unsigned long long first = (unsigned long long)w << kWindowBits;
        const unsigned long long left = n_stored > first ? n_stored - first : 0ull;
        const unsigned nvox = (unsigned)(left < (1ull << kWindowBits) ? left : (1ull << kWindowBits));
        dw_rsrc = __builtin_amdgcn_make_buffer_rsrc(dw + first, 0, (int)(nvox * (unsigned)sizeof(float2)), kRsrcWord3);
        col_rsrc = __builtin_amdgcn_make_buffer_rsrc(COLOR ? crgb + first : crgb, 0, COLOR ? (int)(nvox * (unsigned)sizeof(float4)) : 0, kRsrcWord3);
        w_cur = w;
    };
    set_window(0u);

    auto stage1 = [&](int j, Gather& g /*out: item j*/) {
#if TSDF_INTEGRATE_PRIO & 4
        __builtin_amdgcn_s_setprio(TSDF_INTEGRATE_PRIO_LEVEL);   // as in integrate_kernel
#endif
        const bool have = j < cnt;
        unsigned entry = 0u;                                    // (no item left: entry 0, masked below)
        if (have) {
            const unsigned v = wg_first + (unsigned)wv + NW * (unsigned)j;
            if (__builtin_expect(v >= seg_end, 0)) locate(v);
            entry = v + seg_delta;
        }
        const ItemDesc ds = list[__builtin_amdgcn_readfirstlane(entry)];    // wave-uniform: one scalar 32-byte load
        unsigned long long okm;
        unsigned pixb;
        project_item<KSTD, KTAB>(pc, ds, s_tab, lane, g.pcx, g.pcy, g.pcz, okm, pixb);
        if (!have) okm = 0ull;
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 1) pixb = __builtin_amdgcn_readfirstlane(pixb);       // timing experiment only: one record per wave
#endif
        const unsigned roff = select_by_mask(okm, COLOR ? pixb << 5 : __umul24(pixb, (unsigned)kRec), dropped);
        const unsigned ra = (unsigned)__shfl((int)roff, lane >> 1) + half_off;
        const unsigned rb = (unsigned)__shfl((int)roff, 32 + (lane >> 1)) + half_off;
        if (COLOR) {
            g.A = __builtin_amdgcn_raw_buffer_load_b128(pn_rsrc, (int)ra, 0, 0);
            g.B = __builtin_amdgcn_raw_buffer_load_b128(pn_rsrc, (int)rb, 0, 0);
        } else {
            const u32x3 a3 = __builtin_amdgcn_raw_buffer_load_b96(pn_rsrc, (int)ra, 0, 0);
            const u32x3 b3 = __builtin_amdgcn_raw_buffer_load_b96(pn_rsrc, (int)rb, 0, 0);
            g.A = u32x4{a3.x, a3.y, a3.z, 0u}; g.B = u32x4{b3.x, b3.y, b3.z, 0u};
        }
#if TSDF_INTEGRATE_PRIO & 4
        __builtin_amdgcn_s_setprio(0);
#endif
        g.live = okm;
        g.code = ds.code;
        g.pixb = pixb;
    };
    // append the lanes of `mask` to a queue (a stack: the order in which voxels are updated is free): position = fill +
    // number of mask lanes below this one
    auto push = [&](unsigned long long mask, Entry* q, unsigned head, unsigned& fill, const Entry& e) {
        const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
#if TSDF_QUEUE_FIFO
        if (__builtin_amdgcn_inverse_ballot_w64(mask)) q[(head + fill + pre) & (unsigned)(kQCap - 1)] = e;
#else
        (void)head;
        if (__builtin_amdgcn_inverse_ballot_w64(mask)) q[fill + pre] = e;
#endif
        fill += (unsigned)__popcll(mask);
    };
    // pop the top (up to 64) entries of a queue into the lanes of a batch
    auto pop = [&](const Entry* q, unsigned& head, unsigned& fill, unsigned long long& valid) -> Entry {
        const unsigned n = fill < 64u ? fill : 64u;
        valid = n == 64u ? ~0ull : ((1ull << n) - 1ull);
        fill -= n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // other lanes' queue writes before these reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if TSDF_QUEUE_FIFO
        const Entry e = q[(head + (unsigned)lane) & (unsigned)(kQCap - 1)];   // (lanes >= n read stale entries: masked by `valid`)
        head = (head + n) & (unsigned)(kQCap - 1);
        return e;
#else
        (void)head;
        return q[fill + (unsigned)lane];                           // (lanes >= n read stale entries: masked by `valid`)
#endif
    };
    // Vector-memory operations of the batch stages are issued UNCONDITIONALLY, whether there is a batch or not: a wavefront
    // waits by COUNTING its outstanding operations (s_waitcnt vmcnt(N): all but the N youngest), and hipcc can only
    // count what every path issues.  With the batch's loads and stores inside `if (batch)` it had to assume the smallest
    // count at every join, and the pixel records of S2 then waited for the volume data requested just before them (measured:
    // wavefronts parked 47 % of their life instead of 23 %, 147 us against 136).  Lanes without a voxel (all of them when no
    // queue has 64 entries) carry an out-of-range buffer offset: no memory access, but the operation counts.
    // form a batch from whichever queue has reached `thresh` entries (none: an empty batch) and request its volume data
    auto form_batch = [&](Batch& b, unsigned thresh) {
        Entry e;
        e.x = 0u; e.y = 0u;
        if constexpr (COLOR) { e.z = 0u; e.w = 0u; }
        b.valid = 0ull;
        b.band = false;
        if (fill_p >= thresh) e = pop(q_plain, head_p, fill_p, b.valid);
        else if (fill_b >= thresh) { e = pop(q_band, head_b, fill_b, b.valid); b.band = true; }
        b.off8 = select_by_uniform_mask(b.valid, e.x << 3, kDroppedVolumeOffset);
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 64) b.off8 = select_by_uniform_mask(b.valid, ((unsigned)(blockIdx.x & 255) * 4096u + (unsigned)wv * 64u + (unsigned)lane) << 3, kDroppedVolumeOffset);   // timing experiment only: cache-resident volume accesses
        if (p.debug & 2) b.off8 = kDroppedVolumeOffset;                      // timing experiment only: no volume RMW
#endif
        b.d_new = __uint_as_float(e.y);
        if constexpr (COLOR) {
            b.wc = __uint_as_float(e.z); b.rgb = e.w;
            // band lanes carry their pixel: the f64 cosine of sdf.cpp:294 comes from pack_kernel's plane
            const unsigned off = select_by_uniform_mask(b.band ? b.valid : 0ull, e.z << 3, dropped);
            b.cosb = __builtin_amdgcn_raw_buffer_load_b64(cos_rsrc, (int)off, 0, 0);
        }
        b.old = __builtin_amdgcn_raw_buffer_load_b64(dw_rsrc, (int)b.off8, 0, 0);   // {D,W}: the tracker re-reads these lines -> keep them cached
        if constexpr (COLOR)       // colour is streamed once per frame and never read by the tracker: non-temporal
            b.col = __builtin_amdgcn_raw_buffer_load_b128(col_rsrc, (int)(b.off8 << 1), 0, 2);
    };
    // the running averages of sdf.cpp:289-304 for a batch whose volume data has arrived, and the stores
    auto finish_batch = [&](const Batch& b) {
        u32x2 o2 = u32x2{0u, 0u};
        u32x4 c4 = u32x4{0u, 0u, 0u, 0u};
        if (b.valid != 0ull) {
            float w_new = 1.0f, wc = COLOR ? b.wc : 0.f;                    // sdf.cpp:276; colour weight of a weight-1 voxel = the record's (float)cosine
            if (b.band) {
                w_new = band_weight<EXPPOLY>(b.d_new, eps);                 // sdf.cpp:277-279
                if constexpr (COLOR)                                       // sdf.cpp:295: (float)(w_new * cosine)
                    wc = (float)((double)w_new * __hiloint2double((int)b.cosb.y, (int)b.cosb.x));
            }
            // sdf.cpp:289-292 (D, W) and :294-304 (colour), as packed f32 pairs {D-average, R} and {G, B}
            const float W = __uint_as_float(b.old.y), D = __uint_as_float(b.old.x);
            const float cx = __uint_as_float(b.col.x);
            v2f sum1, num1, num2 = v2f{0.f, 0.f};
            sum1.x = W + w_new;
            num1.x = W * D + w_new * b.d_new;
            if (COLOR) {
                const unsigned rgb = b.rgb;
                const float pr = (float)(rgb & 255u), pg = (float)((rgb >> 8) & 255u), pb = (float)((rgb >> 16) & 255u);
                sum1.y = cx + wc;
                num1.y = cx * __uint_as_float(b.col.y) + wc * pr;
                num2 = v2f{cx, cx} * v2f{__uint_as_float(b.col.z), __uint_as_float(b.col.w)} + v2f{wc, wc} * v2f{pg, pb};
            } else {
                sum1.y = 1.0f; num1.y = 0.0f;
            }
            const v2f r = rcp_refined(sum1);
            v2f q1 = div_core(num1, sum1, r), q2 = v2f{0.f, 0.f};
            if (COLOR) q2 = div_core(num2, v2f{sum1.y, sum1.y}, v2f{r.y, r.y});
            unsigned long long bad = tiny_lanes(num1) | lanes(b.old.y >= kBits2p64);
            if (COLOR) bad |= tiny_lanes(num2) | lanes(b.col.x >= kBits2p64);
            if (__builtin_expect((bad & b.valid) != 0ull, 0)) {
                q1.x = num1.x / sum1.x;
                if (COLOR) { q1.y = num1.y / sum1.y; q2.x = num2.x / sum1.y; q2.y = num2.y / sum1.y; }
            }
            o2.x = __float_as_uint(q1.x); o2.y = __float_as_uint(sum1.x);
            if (COLOR) { c4.x = __float_as_uint(sum1.y); c4.y = __float_as_uint(q1.y); c4.z = __float_as_uint(q2.x); c4.w = __float_as_uint(q2.y); }
        }
        unsigned st8 = b.off8;
#if TSDF_INTEGRATE_DEBUG
        if (p.debug & 32) st8 = kDroppedVolumeOffset;                       // timing experiment only: no stores
#endif
        __builtin_amdgcn_raw_buffer_store_b64(o2, dw_rsrc, (int)st8, 0, 0);        // (lanes without a voxel: dropped)
        if constexpr (COLOR) __builtin_amdgcn_raw_buffer_store_b128(c4, col_rsrc, (int)(st8 << 1), 0, 2);   // nt: colour is streamed
    };
    // everything queued goes out (partial batches): at the end, and before the window of the volume changes.
    // `pending`: the batch whose volume data is in flight; `done`: the other one, already stored.
    auto flush_queues = [&](Batch& pending, Batch& done) {
        finish_batch(pending);
#pragma unroll 1
        while ((fill_p | fill_b) != 0u) { form_batch(done, 1u); finish_batch(done); }
        pending.valid = 0ull; pending.band = false; pending.off8 = kDroppedVolumeOffset;
        done.valid = 0ull; done.band = false; done.off8 = kDroppedVolumeOffset;
    };
    auto stage2 = [&](const Gather& gin /*item j-1, record arrived*/, Batch& pending, Batch& done) {
        u32x4* stage = s_pieces[wv];
        if (COLOR) { stage[lane] = gin.A; stage[64 + lane] = gin.B; }
        else {
            *reinterpret_cast<u32x3*>(&stage[lane]) = u32x3{gin.A.x, gin.A.y, gin.A.z};
            *reinterpret_cast<u32x3*>(&stage[64 + lane]) = u32x3{gin.B.x, gin.B.y, gin.B.z};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x4 P, N;                                            // own record: {Px,Py,Pz,rgb} {Nx,Ny,Nz,(float)cosine}
        if (COLOR) { P = stage[2 * lane + 0]; N = stage[2 * lane + 1]; }
        else {
            const u32x3 p3 = *reinterpret_cast<const u32x3*>(&stage[2 * lane + 0]), n3 = *reinterpret_cast<const u32x3*>(&stage[2 * lane + 1]);
            P = u32x4{p3.x, p3.y, p3.z, 0u}; N = u32x4{n3.x, n3.y, n3.z, 0u};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads above stay before the next step's writes
        __builtin_amdgcn_wave_barrier();
        const float Px = __uint_as_float(P.x), Py = __uint_as_float(P.y), Pz = __uint_as_float(P.z);
        const float Nx = __uint_as_float(N.x), Ny = __uint_as_float(N.y), Nz = __uint_as_float(N.z);
        // sdf.cpp:260: NaN in P.x, P.y or the normal
        const unsigned long long nanm = lanes(__builtin_isunordered(Px, Py)) | lanes(__builtin_isunordered(Nx, Ny)) | lanes(is_nan(Nz));
        // projectivePointToPlaneDistance, sdf.h:177-181 (Eigen dot: a0*b0 + (a1*b1 + a2*b2))
        const double dx = (double)Px - gin.pcx, dy = (double)Py - gin.pcy, dz = (double)Pz - gin.pcz;
        const double p2p = dx * (double)Nx + (dy * (double)Ny + dz * (double)Nz);
        float d = (float)p2p;                                               // sdf.cpp:274
        const unsigned long long okm = gin.live & ~nanm & ~lanes(d > delta);   // sdf.cpp:280-283
        const unsigned long long bandm = okm & lanes(d >= eps);             // sdf.cpp:277-279 (d <= delta holds in okm)
        const unsigned long long plainm = okm & ~bandm;
        // update counts (wave-uniform): rows of the owned x layers / of the halo
        const unsigned row = gin.code >> 6;
        const bool owned = row >= own_row0 && row < own_row1;
        const unsigned n_live = (unsigned)__popcll(okm);
        n_own += owned ? n_live : 0u;
        n_halo += owned ? 0u : n_live;
        if (okm == 0ull) return;                                            // nothing of this item is updated
        d = d < neg_delta ? neg_delta : d;                                  // sdf.cpp:285-287 (band lanes: d >= eps, unchanged)
        // the item's 64 voxels: index in the stored volume = row * m + chunk * 64 + lane; window and window-relative index
        const unsigned long long vox0 = (unsigned long long)row * (unsigned)m + (gin.code & 63u) * 64u;
        if (multi_window) {
            const unsigned w_item = (unsigned)(vox0 >> kWindowBits);
            if (__builtin_expect(w_item != w_cur, 0)) { flush_queues(pending, done); set_window(w_item); }
        }
        Entry e;
        e.x = ((unsigned)vox0 & ((1u << kWindowBits) - 1u)) + (unsigned)lane;
        e.y = __float_as_uint(d);
        if constexpr (COLOR) {
            // sdf.cpp:294-299: weight-1 lanes take the pre-rounded cosine of the record as their colour weight; band lanes
            // carry their pixel and get the f64 cosine when their batch is formed
            e.z = select_by_mask(bandm, gin.pixb, N.w);
            e.w = P.w;
        }
#if TSDF_QUEUE_ALIGN > 1
        // Compaction in ALIGNED groups of TSDF_QUEUE_ALIGN lanes: a group with an updated voxel is queued whole (its other
        // lanes as holes that touch no memory), so that a batch's lane groups stay aligned with the 32 / 64-byte sectors
        // the vector L1 works in (lane-granular compaction costs ~13 more L1 accesses per item: a group of 4 lanes of a
        // colour access then straddles two sectors three times out of four).
        {
            constexpr int GA = TSDF_QUEUE_ALIGN;
            auto widen = [](unsigned long long mk) {
                unsigned long long g = mk;
#pragma unroll
                for (int sft = 1; sft < GA; sft <<= 1) g |= g >> sft;
                constexpr unsigned long long first = GA == 4 ? 0x1111111111111111ull : (GA == 8 ? 0x0101010101010101ull : (GA == 16 ? 0x0001000100010001ull : 1ull));
                g &= first;
                return GA == 64 ? (g ? ~0ull : 0ull) : g * ((1ull << (GA & 63)) - 1ull);
            };
            const unsigned long long wide_p = widen(plainm), wide_b = widen(bandm);
            Entry ep = e, eb = e;
            unsigned hole = kHoleVoxel;
            asm volatile("" : "+v"(hole));
            ep.x = select_by_mask(plainm, e.x, hole);
            eb.x = select_by_mask(bandm, e.x, hole);
            push(wide_p, q_plain, head_p, fill_p, ep);
            if (bandm != 0ull) push(wide_b, q_band, head_b, fill_b, eb);
            return;
        }
#endif
        push(plainm, q_plain, head_p, fill_p, e);
        if (bandm != 0ull) push(bandm, q_band, head_b, fill_b, e);
    };

    Gather G[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        G[q].live = 0ull; G[q].code = 0u; G[q].pixb = 0u;
        G[q].pcx = G[q].pcy = G[q].pcz = 0.0;
        G[q].A = u32x4{0u, 0u, 0u, 0u}; G[q].B = G[q].A;
    }
    // One step = S1(j) request the pixel records of item j | S2(j-1) distance test, queue the updated lanes of item j-1 |
    // form the next batch when a queue has 64 entries and request its volume data | finish and store the batch formed a
    // step ago.  Per step and wavefront: 2 record gathers, 1 cosine gather, 2 volume loads, 2 stores, always in this order.
    // (both queues reaching 64 in the same step is rare; its second batch is finished on the spot)
    for (int j = 0; j < cnt + 1; j += 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            stage1(j + q, G[q]);
            stage2(G[q ^ 1], BB[q ^ 1], BB[q]);
            form_batch(BB[q], 64u);
            finish_batch(BB[q ^ 1]);
            if (__builtin_expect(fill_p >= 64u || fill_b >= 64u, 0)) { finish_batch(BB[q]); form_batch(BB[q], 64u); }
        }
    }
    // (j runs to cnt or cnt + 1: the last item's record has been consumed by an S2 inside the loop; the loop ends after an
    // odd step, so BB[1] is the batch in flight)
    // drain: what is left in the queues goes out as partial batches
    flush_queues(BB[1], BB[0]);

    if (wv == 0 && lane == 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(xcd_fb + kFbTicksWord) + xcd, __builtin_amdgcn_s_memrealtime() - loop_t0 + 1ull);
    __shared__ unsigned s_cnt[2][kIntegrateBlock / 64];
    if (lane == 0) { s_cnt[0][tid >> 6] = n_own; s_cnt[1][tid >> 6] = n_halo; }
    __syncthreads();
    if (tid == 0) {
        unsigned a = 0, b = 0;
        for (int q = 0; q < kIntegrateBlock / 64; ++q) { a += s_cnt[0][q]; b += s_cnt[1][q]; }
        if (a) counters[2 * blockIdx.x + 0] += (unsigned long long)a;
        if (b) counters[2 * blockIdx.x + 1] += (unsigned long long)b;
    }
}

size_t integrate_worklist_entries(const Grid& g) {
    return (size_t)(g.xe - g.xs) * g.m * ((g.m + 63) / 64);
}
// The band regions in front of the overflow region hold at most a quarter of all possible items (a frame lists a few
// percent of them: 9 % at 512^3; whatever the bands' capacities cannot take goes to the overflow region, which is
// integrated like the rest, only without the bands' locality) -- small volumes keep room for everything.
size_t integrate_band_region_entries(const Grid& g) {
    const size_t all = integrate_worklist_entries(g);
    const size_t quarter = all / 4, floor_entries = (size_t)1 << 16;
    return all <= floor_entries ? all : (quarter > floor_entries ? quarter : floor_entries);
}
// band regions + an overflow region that can hold every item; zero-filled once at creation (a wavefront without items
// reads entry 0)
size_t integrate_worklist_bytes(const Grid& g) {
    return (integrate_band_region_entries(g) + integrate_worklist_entries(g) + 8) * sizeof(ItemDesc);
}

int integrate_blocks_per_cu(bool queue) {
    int n = 0;
    const hipError_t e = queue
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, integrate_queue_kernel<true, true, true, true>, kIntegrateBlock, 512 * 24)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, integrate_kernel<true, true, true, true>, kIntegrateBlock, 512 * 24);
    if (e != hipSuccess || n < 1) n = TSDF_INTEGRATE_MIN_WAVES;
    return n;
}

// the queue kernel addresses the volume in windows of 2^27 voxels: any size the row code can express fits
bool integrate_queue_fits(const Grid& g) { return (long long)(g.xe - g.xs) * g.m < (1ll << 26); }

size_t integrate_bookkeeping_words() { return 2 * (size_t)kBinSetWords + kFbWords; }
static_assert((2 * kBinSetWords) % 2 == 0, "the 64-bit tick sums of the feedback block must be 8-byte aligned");

static bool make_tiling(const IntegrateParams& p, IntegrateTiling& tl) {
    const int m = p.g.m;
    const int nx = p.g.xe - p.g.xs;
    tl.n_rows = (long long)nx * m;
    tl.log2m = -1;
    for (int b = 0; b < 31; ++b) if ((1 << b) == m) tl.log2m = b;
    tl.clip = (p.K[6] == 0.0 && p.K[7] == 0.0 && p.K[8] > 0.0) ? 1 : 0;
    tl.k_std = (p.K[1] == 0.0 && p.K[3] == 0.0 && p.K[6] == 0.0 && p.K[7] == 0.0 && p.K[8] == 1.0) ? 1 : 0;
    if (p.debug & 4) tl.k_std = 0;
    // the fixed-point quotients need (dim + 1) << 20 in 32 bits and K * 2^20 finite; anything else divides
    tl.fastq = (p.width <= kMaxFastDim && p.height <= kMaxFastDim) ? 1 : 0;
    for (int a = 0; a < 6; ++a) if (!(fabs(p.K[a]) < 1.0e200)) tl.fastq = 0;
    if (p.debug & 8) tl.fastq = 0;
    return tl.n_rows < (1ll << 26);                                    // row index must fit the 26-bit item code
}
// weight exponent x = -(d-eps)^2/2 with eps <= d <= delta: the Taylor path is valid while |x| <= 0.04
static bool use_exp_poly(const IntegrateParams& p) {
    const double span = (double)p.g.delta - (double)p.g.epsilon;
    return span >= 0.0 && 0.5 * span * span <= 0.04;
}

// The integrate launch is two kernels: the list (launch_integrate_list: list_rows_kernel, whose appended workgroups may
// pack a frame's pixel records) and the items (launch_integrate_items: integrate_kernel over that list).  The list
// depends on the pose and the image SIZE only, so a caller that still waits for the frame's normals can launch it ahead
// (tsdf_integrate_aos); launch_integrate issues both back to back.
hipError_t launch_integrate_list(hipStream_t s, const IntegrateParams& p, void* worklist, unsigned* work_count,
                                 unsigned launch_parity, const PackArgs* pack) {
    const int m = p.g.m;
    const int nx = p.g.xe - p.g.xs;
    if (nx <= 0 || m <= 0) return hipSuccess;
    IntegrateTiling tl;
    if (!make_tiling(p, tl)) return hipErrorInvalidValue;
    // two bookkeeping sets used alternately (see kBinSetWords)
    unsigned* const cur = work_count + (launch_parity & 1) * kBinSetWords;
    unsigned* const xcd_fb = work_count + 2 * kBinSetWords;
    ItemDesc* const list = static_cast<ItemDesc*>(worklist);
    const long long cblocks = (tl.n_rows + kClipBlock - 1) / kClipBlock;
    const unsigned ovf_base = (unsigned)integrate_band_region_entries(p.g);    // band regions in front of the overflow region
    // (pack: the frame's pixel records are still to be written -- workgroups behind the list's own do it, see the kernel)
    const PackArgs no_pack{};
    const unsigned ptiles = pack ? (unsigned)pack_tiles(*pack) : 0u;
    list_rows_kernel<<<dim3((unsigned)cblocks + ptiles), dim3(kClipBlock), 0, s>>>(p, tl, cur, list, ovf_base, xcd_fb,
                                                                                   (unsigned)cblocks, pack ? *pack : no_pack);
    return hipGetLastError();
}

hipError_t launch_integrate_items(hipStream_t s, const IntegrateParams& p, float2* dw, float4* crgb,
                                  const float4* pn, unsigned long long* counters,
                                  void* worklist, unsigned* work_count, int n_blocks,
                                  unsigned launch_parity, unsigned long long* wg_counts, bool queue, const ReleaseWord* release) {
    const int m = p.g.m;
    const int nx = p.g.xe - p.g.xs;
    if (nx <= 0 || m <= 0) return hipSuccess;
    IntegrateTiling tl;
    if (!make_tiling(p, tl)) return hipErrorInvalidValue;
    unsigned* const cur = work_count + (launch_parity & 1) * kBinSetWords;
    unsigned* const nxt = work_count + ((launch_parity + 1) & 1) * kBinSetWords;
    unsigned* const xcd_fb = work_count + 2 * kBinSetWords;
    if (n_blocks < 8 || (n_blocks & 7)) return hipErrorInvalidValue;      // eight XCDs take equal numbers of workgroups
    ItemDesc* const list = static_cast<ItemDesc*>(worklist);
    const unsigned ovf_base = (unsigned)integrate_band_region_entries(p.g);
    const bool exp_poly = use_exp_poly(p);
    const bool ktab = m <= 1024;                                       // 24 bytes of LDS per k
    const size_t lds = ktab ? (size_t)m * 24 : 0;
    const char* planes = reinterpret_cast<const char*>(pn);
    const ReleaseWord rel = release ? *release : ReleaseWord{};
    if (queue && !integrate_queue_fits(p.g)) return hipErrorInvalidValue;
#define TSDF_LAUNCH_INTEGRATE(C, KS, EP, KT) do { \
    if (queue) integrate_queue_kernel<C, KS, EP, KT><<<dim3(n_blocks), dim3(kIntegrateBlock), lds, s>>>(p, tl, list, cur, nxt, ovf_base, counters, dw, crgb, planes, wg_counts, xcd_fb, rel); \
    else integrate_kernel<C, KS, EP, KT><<<dim3(n_blocks), dim3(kIntegrateBlock), lds, s>>>(p, tl, list, cur, nxt, ovf_base, counters, dw, crgb, planes, wg_counts, xcd_fb, rel); } while (0)
#define TSDF_LAUNCH_INTEGRATE3(C, KS, EP) do { if (ktab) TSDF_LAUNCH_INTEGRATE(C, KS, EP, true); else TSDF_LAUNCH_INTEGRATE(C, KS, EP, false); } while (0)
#define TSDF_LAUNCH_INTEGRATE2(C, KS) do { if (exp_poly) TSDF_LAUNCH_INTEGRATE3(C, KS, true); else TSDF_LAUNCH_INTEGRATE3(C, KS, false); } while (0)
    if (p.with_color) { if (tl.k_std) TSDF_LAUNCH_INTEGRATE2(true, true); else TSDF_LAUNCH_INTEGRATE2(true, false); }
    else { if (tl.k_std) TSDF_LAUNCH_INTEGRATE2(false, true); else TSDF_LAUNCH_INTEGRATE2(false, false); }
#undef TSDF_LAUNCH_INTEGRATE2
#undef TSDF_LAUNCH_INTEGRATE3
#undef TSDF_LAUNCH_INTEGRATE
    return hipGetLastError();
}

hipError_t launch_integrate(hipStream_t s, const IntegrateParams& p, float2* dw, float4* crgb,
                            const float4* pn, unsigned long long* counters,
                            void* worklist, unsigned* work_count, int n_blocks,
                            unsigned launch_parity, unsigned long long* wg_counts, bool queue, const PackArgs* pack,
                            const ReleaseWord* release) {
    if (n_blocks < 8 || (n_blocks & 7)) return hipErrorInvalidValue;
    const hipError_t e = launch_integrate_list(s, p, worklist, work_count, launch_parity, pack);
    if (e != hipSuccess) return e;
    return launch_integrate_items(s, p, dw, crgb, pn, counters, worklist, work_count, n_blocks, launch_parity, wg_counts, queue, release);
}

// ------------------------------------------------------------------------------------------------
// SDF::interpolate_distance (sdf.cpp:127-163) on the device layout.
// Returns false when no corner is valid (reference: is_interpolated = false, value NaN).
// `viol` is raised when a corner lies inside the grid but outside this rank's stored layers.

struct Vol {
    const float2* dw;
    int m, xs, xe;
    __device__ __forceinline__ long long dummy() const { return (long long)(xe - xs) * m * m; }   // the {0,0} pair behind the volume
};

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
    const long long base = (((long long)bi - V.xs) * m + bj) * m + kc;
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
    Vol V{dw, g.m, g.xs, g.xe};
    float out = 0.0f;
    unsigned viol = 0;
    const bool ok = interp(V, vox[3 * t + 0], vox[3 * t + 1], vox[3 * t + 2], out, viol);
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
    unsigned long long* stamps = nullptr;   // diagnosis (TSDF_TRACK_STAMPS=1): 8 words per workgroup, s_memrealtime at the phase boundaries
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

__global__ __launch_bounds__(kTrackBlock) void track_kernel(TrackParams p, const float2* __restrict__ dw,
                                                             const float4* __restrict__ samples,
                                                             double* __restrict__ partials, TrackFold fold) {
    constexpr int NW = kTrackBlock / 64;
    __shared__ unsigned long long s_in[1], s_oog[1];       // the 64-sample window of this workgroup
    __shared__ unsigned long long s_in2[NW], s_oog2[NW];   // look-ahead windows
    __shared__ double s_red[NW][8][8];                     // [wave][q][slot]
    __shared__ double s_rpm[54];                           // the six perturbed rotations, for lane-indexed access

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = blockIdx.x * kSamplesPerBlock;        // first of this workgroup's own samples
    if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
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

    if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
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

    if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
    // ---- the look-ups of this lane (camera_tracking.cpp:269-361): slot A = look-up q (centre, +x -x +y -y +z -z)
    // on lanes 0..6, slot B = look-up 7 + q (r1p r1m r2p r2m r3p r3m) on lanes 0..5
    const bool owned = (cls == kClsIn) && (sg.vx >= (double)p.g.own_x0) && (sg.vx < (double)p.g.own_x1);
    float valA = 0.0f, valB = 0.0f;
    unsigned viol = 0;
    bool okA = false, okB = false;
    {
        // both look-ups are issued before either is evaluated; lanes without a look-up run the same code on their
        // sample's centre (in-cache, results masked) so that the whole section stays one basic block
        const bool actA = owned && q < 7, actB = owned && q < 6;
        Vol V{dw, p.g.m, p.g.xs, p.g.xe};
        double ax = sg.vx, ay = sg.vy, az = sg.vz;
        if (q >= 1 && q < 7) {
            const int a = (q - 1) >> 1;
            const double step = ((q - 1) & 1) ? -(double)p.v_h : (double)p.v_h;
            ax += (a == 0) ? step : 0.0; ay += (a == 1) ? step : 0.0; az += (a == 2) ? step : 0.0;
        }
        double bx, by, bz;
        voxel_of(p, &s_rpm[9 * (q < 6 ? q : 0)], sg, bx, by, bz);
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

    if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
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

    if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
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
    if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
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
            if (fold.stamps && tid == 0 && blockIdx.x < (unsigned)kTrackStampBlocks) fold.stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
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
                               unsigned long long word, unsigned long long pass, const PeerExchange* peers, unsigned long long* stamps,
                               AqlQueue* aql) {
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
    f.stamps = stamps;
    if (aql) {
        TrackKernarg ka{p, dw, samples, partials, f};
        if (aql->submit(&ka, (uint32_t)nb, (uint32_t)kTrackBlock)) return hipSuccess;
    }
    track_kernel<<<dim3(nb), dim3(kTrackBlock), 0, s>>>(p, dw, samples, partials, f);
    return hipGetLastError();
}
size_t track_kernel_explicit_arg_bytes() { return sizeof(TrackKernarg); }
const char* track_kernel_symbol_prefix() { return "_ZN4tsdf12track_kernelE"; }

// (two sets, used alternately by pass parity: with passes coming off two queues a set's re-zeroing store is no longer
// ordered before the NEXT pass's arrivals, only before the one after it)
size_t track_fold_counter_words() { return 32 * (size_t)(kTrackShards + 1); }
int track_num_shards(int32_t n_samples) {
    const int nb = track_num_blocks(n_samples);
    return nb < kTrackShards ? nb : kTrackShards;
}

// ------------------------------------------------------------------------------------------------
// (de)interleave helpers for tsdf_download / tsdf_upload (reference-order host mirrors)

__global__ __launch_bounds__(256) void split_kernel(const float2* __restrict__ dw, float* __restrict__ d,
                                                     float* __restrict__ w, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = dw[i];
        d[i] = v.x; w[i] = v.y;
    }
}
__global__ __launch_bounds__(256) void merge_kernel(float2* __restrict__ dw, const float* __restrict__ d,
                                                     const float* __restrict__ w, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dw[i] = make_float2(d[i], w[i]);
}
__global__ __launch_bounds__(256) void split4_kernel(const float4* __restrict__ c, float* __restrict__ a,
                                                      float* __restrict__ r, float* __restrict__ g,
                                                      float* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 v = c[i];
        a[i] = v.x; r[i] = v.y; g[i] = v.z; b[i] = v.w;
    }
}
__global__ __launch_bounds__(256) void merge4_kernel(float4* __restrict__ c, const float* __restrict__ a,
                                                      const float* __restrict__ r, const float* __restrict__ g,
                                                      const float* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        c[i] = make_float4(a[i], r[i], g[i], b[i]);
}

static inline unsigned stream_blocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}
hipError_t launch_split(hipStream_t s, const float2* dw, float* d, float* w, int64_t n) {
    split_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(dw, d, w, n);
    return hipGetLastError();
}
hipError_t launch_merge(hipStream_t s, float2* dw, const float* d, const float* w, int64_t n) {
    merge_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(dw, d, w, n);
    return hipGetLastError();
}
hipError_t launch_split4(hipStream_t s, const float4* c, float* a, float* r, float* g, float* b, int64_t n) {
    split4_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(c, a, r, g, b, n);
    return hipGetLastError();
}
hipError_t launch_merge4(hipStream_t s, float4* c, const float* a, const float* r, const float* g, const float* b, int64_t n) {
    merge4_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(c, a, r, g, b, n);
    return hipGetLastError();
}

}  // namespace tsdf
