// integrate_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the TSDF integration pass.
//
//   pack_kernel        per-frame image packing (xyz|nrm|rgb planes -> 32-byte pixel records + the
//                      tracker's column-major stride-3 sample list, camera_tracking.cpp:162-163)
//   list_rows_kernel   per k-row frustum interval -> list of 64-voxel work items, sorted by image band
//   integrate_kernel   SDF::update over that list          (reference src/sdf.cpp:224-315)
//
// Numerics: every operation that decides a result (f64 geometry, f32 running averages, (int) truncations) is the
// reference's operation in the reference's order, so this file MUST be compiled with -ffp-contract=off (no FMA
// contraction) and without fast-math.  HBM/L2-bound byte movers: no MFMA anywhere (there is no dense contraction).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "device_util.h"
#include "tsdf_device.h"

namespace tsdf {

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

constexpr int kClipBlock = 256;                  // rows (threads) per workgroup of list_rows_kernel
constexpr int kIntegrateMinWaves = 5;            // waves per SIMD the register allocator must leave room for (<= 96 VGPRs)

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
constexpr int kBins = 64;
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
__device__ __forceinline__ void update_xcd_shares(unsigned* fb) {
    unsigned long long* ticks = reinterpret_cast<unsigned long long*>(fb + kFbTicksWord);
    if (fb[8] != kFbOne) {                                   // first launch: equal shares
        for (int x = 0; x <= 8; ++x) fb[x] = (unsigned)x * (kFbOne / 8u);
    } else {
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
        const int gi = grid_global_layer(p.g, il);                     // (il + xs for a plain slab)
        // get_global_coordinates, sdf.h:153-157: (extent/(float)m) * (i + 0.5) + origin
        const double gx = cw * ((double)gi + 0.5) + p.g.origin[0];
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
        if (klo <= khi && gi >= 0 && gi < m) { c0 = klo >> 6; n = (khi >> 6) - c0 + 1; }      // (a block-cyclic handle's padding layers hold no voxel)
    }
    unsigned rank = 0u;
    if (n) rank = atomicAdd(&s_wg[bin], (unsigned)n);
    __syncthreads();
    // one returning atomic per band with items: the group's place in the band's region, or in the overflow region
    for (int t = tid; t < kBins; t += kClipBlock) {
        const unsigned cnt = s_wg[t];
        if (cnt) {
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
        const double gx = (double)p.g.cell_w * ((double)grid_global_layer(p.g, il) + 0.5) + p.g.origin[0];
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
//     vector-memory instruction per item).

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kFixShift = 20;                        // pixel coordinates in 2^-20 units
constexpr unsigned kFixOne = 1u << kFixShift;
constexpr int kMaxFastDim = 2047;                    // (dim + 1) << 20 must fit 32 bits
constexpr unsigned kDroppedOffset = 0x7fffffffu;     // beyond every buffer: the lane loads zeros / stores nothing
constexpr int kRsrcWord3 = 0x00020000;               // raw buffer, 32-bit data format (gfx9 family)

// Per-pixel data of a frame, written by pack_kernel into ONE buffer of kPixelRecordBytes per pixel (record index
// rec = col*pix_su + row*pix_sv):  with colour 32-byte records {Px,Py,Pz,rgb}{Nx,Ny,Nz,(float)cosine};  without colour
// 24-byte records {Px,Py,Pz,Nx,Ny,Nz}.
static_assert(kPixelRecordBytes == 32, "two float4 per pixel");

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

                                    // instructions, bit 1 = around the volume loads (measurement builds)

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

template <bool COLOR, bool KSTD, bool EXPPOLY, bool KTAB, bool CYC /* block-cyclic placement: which rows are the handle's own */>
__global__ __launch_bounds__(kIntegrateBlock, kIntegrateMinWaves) void integrate_kernel(
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
    // ... requested one item ahead: stage 1 used to open with the load and an s_waitcnt lgkmcnt(0) right behind it -- a
    // trip to L2 at the stage's raised priority in front of every item
    ItemDesc dnext = fetch_desc(item_v(0));
    auto stage1 = [&](int j, GatherState& g /*out: item j*/) {
        // Stage 1 runs at raised wave priority: it ends in the gathers, the longest trip of an item (64 scattered records
        // through L1 / L2), and a wavefront on its way to them should not queue behind the arithmetic of its four
        // neighbours on the SIMD.  Measured on five boxes, alternating builds: with the priority only around the two gather
        // instructions integrate_kernel is 2-7 % shorter on four of them (113.5 -> 105.7-110.4 us, 113.7 -> 110.2-111.0,
        // 113.4 -> 110.4, 108.9 -> 106.4) and sits on two levels (106.3 / 110.5 against 108.7) on the fifth; the whole
        // stage takes another 1.0-1.2 us (109.1-109.5); stage 2 at the SAME priority makes it worse, stage 2 one step above
        // stage 3 (3 / 1 / 0) another ~3 us (109.3 -> 106.4 in 5 of 5 alternations on one box; 105.0 -> 101.2-102.1 in two
        // of three on another, 106.1 in the third).
        __builtin_amdgcn_s_setprio(3);
        const bool have = j < cnt;
        const ItemDesc ds = dnext;                              // requested at the end of the previous item's stage 1
        unsigned long long okm;
        unsigned pixb;
        project_item<KSTD, KTAB>(pc, ds, s_tab, lane, g.pcx, g.pcy, g.pcz, okm, pixb);
        if (!have) okm = 0ull;
        // Pixel-record gather, paired: the vector L1 serves a wave's gather about one lane-address at a time, and the
        // two halves of a record are two instructions.  Instead the first load fetches both halves of
        // the records of lanes 0..31 (lane l: record of lane l/2, half l%2), the second those of lanes 32..63: lane pairs
        // share a line, so the look-ups of an item are halved.  Stage 2 un-shuffles the pieces through a wave-private
        // LDS buffer.  Dead lanes carry an offset beyond the plane: no look-up at all.
        const unsigned roff = select_by_mask(okm, COLOR ? pixb << 5 : __umul24(pixb, (unsigned)kRec), dropped);
        const unsigned ra = (unsigned)__shfl((int)roff, lane >> 1) + half_off;
        const unsigned rb = (unsigned)__shfl((int)roff, 32 + (lane >> 1)) + half_off;
        if (COLOR) {
            g.A = __builtin_amdgcn_raw_buffer_load_b128(pn_rsrc, (int)ra, 0, 0);      // piece for LDS slot lane
            g.B = __builtin_amdgcn_raw_buffer_load_b128(pn_rsrc, (int)rb, 0, 0);      // piece for LDS slot 64 + lane
        } else {
            const u32x3 a3 = __builtin_amdgcn_raw_buffer_load_b96(pn_rsrc, (int)ra, 0, 0);
            const u32x3 b3 = __builtin_amdgcn_raw_buffer_load_b96(pn_rsrc, (int)rb, 0, 0);
            g.A = u32x4{a3.x, a3.y, a3.z, 0u}; g.B = u32x4{b3.x, b3.y, b3.z, 0u};
        }
        dnext = fetch_desc(item_v(j + 1));
        __builtin_amdgcn_s_setprio(0);
        g.live = okm;
        g.code = ds.code;
    };
    auto stage2 = [&](const GatherState& gin /*item j-1, record arrived*/, UpdateState& u /*out: item j-1*/) {
        __builtin_amdgcn_s_setprio(1);
        float d = 0.f, wn = 1.0f, wc = 0.f;
        unsigned rgbv = 0u;
        unsigned long long okm = 0ull;
        // An item none of whose lanes projects into the image skips the un-shuffle and the distances; like stage 3's skip
        // this leaves the vector-memory operations of a step where they are.  Putting the volume loads or the stores of
        // items that update nothing behind the same kind of branch costs 3-10 us: hipcc then waits for the smallest
        // outstanding count at every use (r04_integrate_fixed_costs.json).
        if (gin.live != 0ull)
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
        // sdf.cpp:294-299: wc = (float)(w_new * cosine).  For w_new == 1 that is the pre-rounded cosine of the record;
        // a wavefront with lanes in the exp() band recomputes the f64 cosine from the normal for those lanes.
        wc = COLOR ? __uint_as_float(N.w) : 0.f;
        if (bandm != 0ull) {
            wn = __uint_as_float(select_by_mask(bandm, __float_as_uint(band_weight<EXPPOLY>(d, eps)), 0x3f800000u));
            if (COLOR) {
                const double nxd = (double)Nx, nyd = (double)Ny, nzd = (double)Nz;
                const double n2 = nxd * nxd + (nyd * nyd + nzd * nzd);
                const unsigned long long plain = lanes(n2 >= 0x1p-200) & lanes(n2 <= 0x1p200);
                double cosine;
                if (__builtin_expect((bandm & ~plain) == 0ull, 1)) cosine = pixel_cosine_core(nzd, n2);
                else cosine = pixel_cosine(Nx, Ny, Nz);
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
        u.off8 = select_by_mask(okm, lane8, dropped);
        const __amdgpu_buffer_rsrc_t seg_dw = __builtin_amdgcn_make_buffer_rsrc(dw + base2, 0, 64 * (int)sizeof(float2), kRsrcWord3);
        const unsigned ld8 = u.off8;
        u.old = __builtin_amdgcn_raw_buffer_load_b64(seg_dw, (int)ld8, 0, 0);   // {D,W}: the tracker re-reads these lines -> keep them cached
        if (COLOR) {   // colour is streamed once per frame and never read by the tracker: non-temporal
            const __amdgpu_buffer_rsrc_t seg_c = __builtin_amdgcn_make_buffer_rsrc(crgb + base2, 0, 64 * (int)sizeof(float4), kRsrcWord3);
            u.col = __builtin_amdgcn_raw_buffer_load_b128(seg_c, (int)(ld8 << 1), 0, 2);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto stage3 = [&](const UpdateState& uin /*item j-1-DEPTH, volume data arrived*/) {
        const unsigned code3 = uin.code;
        const unsigned row3 = code3 >> 6;
        bool owned3;                                                        // wave-uniform; rows of the owned x layers
        if (!CYC) owned3 = row3 >= own_row0 && row3 < own_row1;
        else {      // local layer -> position in its block (m is a power of two here; blocks lie inside the grid: tsdf_create)
            const unsigned il3 = row3 >> (unsigned)tl.log2m;
            const unsigned o3 = il3 - __umulhi(il3, p.g.blk_magic) * (unsigned)p.g.blk_layers;
            owned3 = o3 - (unsigned)(p.g.own_x0 - p.g.blk_first) < (unsigned)p.g.blk_own;
        }
        long long base3 = (long long)row3 * m + (long long)(code3 & 63u) * 64;
        const unsigned n_live = (unsigned)__popcll(uin.live);
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
    constexpr int kDepth = 1;
    constexpr int NU = kDepth + 1, NG = 2;
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
    // The pipeline's fill and drain, peeled: the rolled loop below runs cnt + 2 (or 3) steps of three stages each, i.e.
    // 2-3 steps' worth of stages on items that do not exist -- with ~38 items per wavefront that is 6 % of all issued
    // instructions.  While the kernel's stages queued behind each other's arithmetic that changed nothing (measured in
    // the round's first half); with the graded priorities the kernel sits within ~10 % of its issue floor and the peeled
    // form is worth 0.8 us (105.3 -> 104.5 us, 4 of 4 alternations).  The steady part is the same two-step rotation.
    static_assert(kDepth == 1, "the peeled pipeline is written for two rotating update states");
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
    for (int j = 0; j < cnt + 1 + kDepth; j += PERIOD) {
#pragma unroll
        for (int q = 0; q < PERIOD; ++q) {
            // S1(j+q) -> G[q%2];  S2(j+q-1): G[(q+1)%2] -> U[q%NU];  S3(j+q-1-DEPTH): U[(q+1)%NU] (the oldest)
            stage1(j + q, G[q % NG]);
            stage2(G[(q + 1) % NG], U[q % NU]);
            stage3(U[(q + 1) % NU]);
        }
    }
    // (steps run up to j >= cnt + DEPTH, so S3 has retired item cnt-1 inside the loop: nothing to drain)

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
size_t integrate_worklist_entries(const Grid& g) {
    return (size_t)grid_stored_layers(g) * g.m * ((g.m + 63) / 64);
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

int integrate_blocks_per_cu() {
    int n = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, integrate_kernel<true, true, true, true, false>, kIntegrateBlock, 512 * 24);
    if (e != hipSuccess || n < 1) n = kIntegrateMinWaves;
    return n;
}


size_t integrate_bookkeeping_words() { return 2 * (size_t)kBinSetWords + kFbWords; }
static_assert((2 * kBinSetWords) % 2 == 0, "the 64-bit tick sums of the feedback block must be 8-byte aligned");

static bool make_tiling(const IntegrateParams& p, IntegrateTiling& tl) {
    const int m = p.g.m;
    const int nx = grid_stored_layers(p.g);
    tl.n_rows = (long long)nx * m;
    tl.log2m = -1;
    for (int b = 0; b < 31; ++b) if ((1 << b) == m) tl.log2m = b;
    tl.clip = (p.K[6] == 0.0 && p.K[7] == 0.0 && p.K[8] > 0.0) ? 1 : 0;
    tl.k_std = (p.K[1] == 0.0 && p.K[3] == 0.0 && p.K[6] == 0.0 && p.K[7] == 0.0 && p.K[8] == 1.0) ? 1 : 0;
    // the fixed-point quotients need (dim + 1) << 20 in 32 bits and K * 2^20 finite; anything else divides
    tl.fastq = (p.width <= kMaxFastDim && p.height <= kMaxFastDim) ? 1 : 0;
    for (int a = 0; a < 6; ++a) if (!(fabs(p.K[a]) < 1.0e200)) tl.fastq = 0;
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
    const int nx = grid_stored_layers(p.g);
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
                                  unsigned launch_parity, unsigned long long* wg_counts, const ReleaseWord* release) {
    const int m = p.g.m;
    const int nx = grid_stored_layers(p.g);
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
    const bool cyc = p.g.blk_own > 0;
    const size_t lds = ktab ? (size_t)m * 24 : 0;
    const char* planes = reinterpret_cast<const char*>(pn);
    const ReleaseWord rel = release ? *release : ReleaseWord{};
#define TSDF_LAUNCH_INTEGRATE(C, KS, EP, KT) \
    do { if (cyc) integrate_kernel<C, KS, EP, KT, true><<<dim3(n_blocks), dim3(kIntegrateBlock), lds, s>>>(p, tl, list, cur, nxt, ovf_base, counters, dw, crgb, planes, wg_counts, xcd_fb, rel); \
         else integrate_kernel<C, KS, EP, KT, false><<<dim3(n_blocks), dim3(kIntegrateBlock), lds, s>>>(p, tl, list, cur, nxt, ovf_base, counters, dw, crgb, planes, wg_counts, xcd_fb, rel); } while (0)
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
                            unsigned launch_parity, unsigned long long* wg_counts, const PackArgs* pack,
                            const ReleaseWord* release) {
    if (n_blocks < 8 || (n_blocks & 7)) return hipErrorInvalidValue;
    const hipError_t e = launch_integrate_list(s, p, worklist, work_count, launch_parity, pack);
    if (e != hipSuccess) return e;
    return launch_integrate_items(s, p, dw, crgb, pn, counters, worklist, work_count, n_blocks, launch_parity, wg_counts, release);
}

}  // namespace tsdf
