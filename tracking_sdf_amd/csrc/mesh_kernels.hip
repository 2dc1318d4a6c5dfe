// mesh_kernels.hip -- iso-surface extraction from the TSDF volume: the work of the
// reference's visualiser thread (sdf.cpp:317-391 -> pcl::MarchingCubesSDF::performReconstruction,
// marching_cubes_sdf.cpp:243-287), on the volume where it lives.
//
// What the reference computes, restated for the device layout:
//   * cubes: one per interior voxel (1 <= i,j,k <= m-2, sdf.cpp:36-39) in index order (i slowest, k fastest);
//     cube corners 0..7 = voxel + {0, x, x+z, z, y, x+y, x+y+z, y+z}  (getNeighborList1D, :203-218);
//   * gate: all eight W > 0, else the cube is empty (:219-239 make it degenerate);
//   * case number: bit c set when D[c] < iso (:108-115); crossed edge e gets the vertex
//     p1 + (iso - v1)/(v2 - v1) * (p2 - p1) in float (:87-94), corner positions extent*idx/m and
//     that + extent/m (:121-141) -- the reference's mesh sits half a voxel off the voxel centres, kept;
//   * output: a triangle soup, 3 vertices per triangle, cubes in index order (the per-thread clouds are
//     concatenated in thread order = index order for a static schedule, :262-283).
//   * SDF::visualize (sdf.cpp:353-383) adds sdf_origin in double and colours every vertex with
//     SDF::interpolate_color (sdf.cpp:164-217).
// The triangle table is mc_tables.h: the reference's table (marching_cubes_sdf.h:73-364) kept as data, so the soup is
// performReconstruction's triangle for triangle (tools/gen_mc_tables.py --check verifies it against the cube geometry).
//
// Five launches; the first is the 8 B/voxel D,W sweep and is HBM-bound, the rest touch the surface only:
//   mesh_count_kernel        one wavefront per (i,j) row of cubes: case numbers, triangles per row
//   mesh_scan_groups_kernel  exclusive scan of the row counts inside groups of 1024 rows
//   mesh_scan_top_kernel     exclusive scan of the group sums (one workgroup), total
//   mesh_list_kernel         rows with triangles only: case numbers again, one 8-byte descriptor
//                            (row, k, case, t) per triangle at its final position
//   mesh_vertex_kernel       one thread per output vertex: position (+ colour) from the descriptor --
//                            evenly loaded however the surface is distributed over the rows, and the
//                            12-byte vertex / 16-byte colour stores of a wavefront are consecutive
// Recomputing the case numbers in the list pass costs a second sweep of the rows that have surface
// (a few per cent of the volume) instead of 1 B/voxel of scratch written and read back.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "tsdf_device.h"

namespace tsdf {

#define MC_TABLE_DECL __device__ __attribute__((aligned(16))) const
#include "mc_tables.h"

constexpr int kMeshBlock = 256;

__device__ __forceinline__ int mesh_trunc_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// ---- reading a row of cubes ---------------------------------------------------------------------------
// A wavefront walks one (i,j) row of cubes along k.  Lane l holds the voxels k = kb + 2l and k + 1 of the four
// voxel rows (i,j), (i+1,j), (i,j+1), (i+1,j+1) -- one 16-byte load per row -- and gets voxel k + 2 from lane
// l + 1, so it has the corners of the two cubes k and k + 1.  Lane 63 has no neighbour and only supplies data:
// 126 cubes per step.  (Eight 8-byte loads per cube, the direct way, made the vector L1 the bottleneck.)
typedef float mesh_f4 __attribute__((ext_vector_type(4), aligned(8)));   // rows of an odd m start 8-byte aligned
typedef float mesh_f2 __attribute__((ext_vector_type(2)));
constexpr int kMeshStep = 126;

struct MeshRowPtrs { const float* r[4]; };      // (i,j) (i+1,j) (i,j+1) (i+1,j+1), as floats {D,W} pairs

__device__ __forceinline__ MeshRowPtrs mesh_rows_of(const MeshParams& p, const float2* __restrict__ dw, int i, int j) {
    const long long m = p.g.m;
    const float* base = reinterpret_cast<const float*>(dw + ((long long)(i - p.g.xs) * m + j) * m);
    MeshRowPtrs q;
    q.r[0] = base; q.r[1] = base + 2 * m * m; q.r[2] = base + 2 * m; q.r[3] = base + 2 * m * m + 2 * m;
    return q;
}

// voxels k, k+1 (own load) and k+2 (from lane+1) of one voxel row.  The load and the shuffle are separate so that a
// caller can keep the raw load in flight (the shuffle needs the data and would wait for it on the spot).
struct MeshRow3 { float d[3], w[3]; };
__device__ __forceinline__ mesh_f4 mesh_load_raw(const float* __restrict__ row, int k, int m) {
    mesh_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k + 1 < m) v = *reinterpret_cast<const mesh_f4*>(row + 2 * k);
    else if (k < m) { const mesh_f2 h = *reinterpret_cast<const mesh_f2*>(row + 2 * k); v.x = h.x; v.y = h.y; }
    return v;
}
__device__ __forceinline__ MeshRow3 mesh_expand(const mesh_f4 v) {
    MeshRow3 r;
    r.d[0] = v.x; r.w[0] = v.y; r.d[1] = v.z; r.w[1] = v.w;
    r.d[2] = __shfl_down(v.x, 1, 64); r.w[2] = __shfl_down(v.y, 1, 64);
    return r;
}
__device__ __forceinline__ MeshRow3 mesh_load_row(const float* __restrict__ row, int k, int m) {
    return mesh_expand(mesh_load_raw(row, k, m));
}

// case numbers of the cubes k (c0) and k + 1 (c1) from the voxel rows (i,j) (i+1,j) (i,j+1) (i+1,j+1); 0 when
// the weight gate fails, the cube is not interior, or the lane is the data-only lane 63
__device__ __forceinline__ void mesh_classify(const MeshParams& p, const MeshRow3& r0, const MeshRow3& r1, const MeshRow3& r2,
                                              const MeshRow3& r3, int k, int lane, int& c0, int& c1) {
    const MeshRow3* rows[4] = {&r0, &r1, &r2, &r3};
    // corners 0..7 = (row, k offset): (0,0) (1,0) (1,1) (0,1) (2,0) (3,0) (3,1) (2,1)   (getNeighborList1D :203-218)
    const int crow[8] = {0, 1, 1, 0, 2, 3, 3, 2}, coff[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    int idx[2] = {0, 0};
    bool all[2] = {true, true};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            all[u] = all[u] && (rows[crow[c]]->w[coff[c] + u] > 0.0f);
            if (rows[crow[c]]->d[coff[c] + u] < p.iso) idx[u] |= 1 << c;
        }
    const bool live = lane < 63;
    const int m = p.g.m;
    c0 = (live && all[0] && k >= 1 && k <= m - 2) ? idx[0] : 0;
    c1 = (live && all[1] && k + 1 <= m - 2) ? idx[1] : 0;
}

__device__ __forceinline__ void mesh_two_cubes(const MeshParams& p, const MeshRowPtrs& q, int kb, int lane, int& c0, int& c1) {
    const int m = p.g.m, k = kb + 2 * lane;
    // the four loads first, then the shuffles (a shuffle waits for its load)
    const mesh_f4 v0 = mesh_load_raw(q.r[0], k, m), v1 = mesh_load_raw(q.r[1], k, m);
    const mesh_f4 v2 = mesh_load_raw(q.r[2], k, m), v3 = mesh_load_raw(q.r[3], k, m);
    const MeshRow3 r0 = mesh_expand(v0), r1 = mesh_expand(v1), r2 = mesh_expand(v2), r3 = mesh_expand(v3);
    mesh_classify(p, r0, r1, r2, r3, k, lane, c0, c1);
}

// rows: r = (i - ci0) * (m-2) + (j-1).  A workgroup takes kMeshRowsPerBlock consecutive rows of one layer, one
// wavefront per row: the row (i, j+1) that two neighbouring cube rows share is then fetched by two waves of
// the same CU.  Workgroups are dealt round-robin to the 8 XCDs; every XCD gets one contiguous eighth of the
// row groups so that the layer i+1 it reads for layer i is still in its L2 when it gets to layer i+1.
constexpr int kMeshRowsPerBlock = kMeshBlock / 64;      // 4
__device__ __forceinline__ int mesh_group_of_block(unsigned block, int n_groups) {
    const int per = (n_groups + 7) >> 3;
    return (int)(block & 7u) * per + (int)(block >> 3);
}
__host__ __device__ inline int mesh_groups_per_layer(int m) { return (m - 2 + kMeshRowsPerBlock - 1) / kMeshRowsPerBlock; }
__host__ inline unsigned mesh_grid(const MeshParams& p) {
    const long long n_groups = (long long)(p.ci1 - p.ci0) * mesh_groups_per_layer(p.g.m);
    return (unsigned)(((n_groups + 7) >> 3) << 3);
}
// (layer, first row of the group) of this workgroup; false when the workgroup is padding
__device__ __forceinline__ bool mesh_block_rows(const MeshParams& p, int& i, int& j0) {
    const int gpl = mesh_groups_per_layer(p.g.m);
    const int n_groups = (p.ci1 - p.ci0) * gpl;
    const int grp = mesh_group_of_block(blockIdx.x, n_groups);
    if (grp >= n_groups) return false;
    i = p.ci0 + grp / gpl;
    j0 = 1 + (grp % gpl) * kMeshRowsPerBlock;
    return true;
}

// Count pass.  Unlike the list pass it sweeps the whole volume, so it is laid out for HBM: a wavefront owns
// 126 cubes of one row and MARCHES over kMeshLayerChunk layers, keeping the two voxel rows of layer i+1 in
// registers as the rows of layer i of the next step -- every voxel row is fetched once per wave (twice per CU:
// rows j and j+1 of neighbouring waves) instead of four times.  Rows get their counts by integer atomics
// (row_count is zeroed first; a row has up to ceil(m/126) contributing wavefronts).
#ifndef TSDF_MESH_LAYER_CHUNK
#define TSDF_MESH_LAYER_CHUNK 16
#endif
constexpr int kMeshLayerChunk = TSDF_MESH_LAYER_CHUNK;
__host__ __device__ inline int mesh_ksteps(int m) { return (m - 2) / kMeshStep + 1; }       // kb = 0, 126, ... <= m-2
__host__ inline unsigned mesh_count_grid(const MeshParams& p) {
    const long long chunks = (p.ci1 - p.ci0 + kMeshLayerChunk - 1) / kMeshLayerChunk;
    const long long units = chunks * mesh_groups_per_layer(p.g.m) * mesh_ksteps(p.g.m);
    return (unsigned)(((units + 7) >> 3) << 3);
}
__global__ __launch_bounds__(kMeshBlock) void mesh_count_kernel(MeshParams p, const float2* __restrict__ dw,
                                                                 unsigned* __restrict__ row_count) {
    __shared__ unsigned char s_ntri[256];
    s_ntri[threadIdx.x] = kMcNumTri[threadIdx.x];
    __syncthreads();
    const int m = p.g.m, inner = m - 2;
    const int gpl = mesh_groups_per_layer(m), ksteps = mesh_ksteps(m);
    const int chunks = (p.ci1 - p.ci0 + kMeshLayerChunk - 1) / kMeshLayerChunk;
    const int n_units = chunks * gpl * ksteps;
    const int unit = mesh_group_of_block(blockIdx.x, n_units);
    if (unit >= n_units) return;
    const int s = unit % ksteps, jg = (unit / ksteps) % gpl, chunk = unit / (ksteps * gpl);
    const int lane = threadIdx.x & 63;
    const int j = 1 + jg * kMeshRowsPerBlock + (int)(threadIdx.x >> 6);
    if (j > inner) return;                                   // wave-uniform
    const int i_begin = p.ci0 + chunk * kMeshLayerChunk;
    const int i_end = (i_begin + kMeshLayerChunk < p.ci1) ? i_begin + kMeshLayerChunk : p.ci1;
    const int k = s * kMeshStep + 2 * lane;
    const long long plane = 2ll * m * m;
    const float* row_j = reinterpret_cast<const float*>(dw + ((long long)(i_begin - p.g.xs) * m + j) * m);
    // layers i, i+1 expanded; layer i+2 raw (requested one step ago); layer i+3 requested at the top of the step:
    // every load has two steps to arrive
    const int last_layer = p.g.xe - 1;
    auto layer_ptr = [&](int layer) { return row_j + (long long)((layer < last_layer ? layer : last_layer) - i_begin) * plane; };
    MeshRow3 a0 = mesh_load_row(row_j, k, m), a2 = mesh_load_row(row_j + 2 * m, k, m);
    MeshRow3 b0 = mesh_load_row(layer_ptr(i_begin + 1), k, m), b2 = mesh_load_row(layer_ptr(i_begin + 1) + 2 * m, k, m);
    mesh_f4 c0r = mesh_load_raw(layer_ptr(i_begin + 2), k, m), c2r = mesh_load_raw(layer_ptr(i_begin + 2) + 2 * m, k, m);
    for (int i = i_begin; i < i_end; ++i) {
        const float* nxt = layer_ptr(i + 3);
        const mesh_f4 d0r = mesh_load_raw(nxt, k, m), d2r = mesh_load_raw(nxt + 2 * m, k, m);
        int c0, c1;
        mesh_classify(p, a0, b0, a2, b2, k, lane, c0, c1);
        unsigned n = (unsigned)s_ntri[c0] + (unsigned)s_ntri[c1];
        if (__ballot(n != 0u)) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
            if (lane == 0) atomicAdd(&row_count[(i - p.ci0) * inner + (j - 1)], n);
        }
        a0 = b0; a2 = b2;
        b0 = mesh_expand(c0r); b2 = mesh_expand(c2r);
        c0r = d0r; c2r = d2r;
    }
}

// Exclusive scan of the row counts in two levels: groups of 1024 rows are scanned by one workgroup each
// (row_offset = offset inside the group, group_sum), then one workgroup scans the group sums into group_base
// (<= 4096 groups for 2046^2 rows) and the total.  Row r starts at group_base[r >> 10] + row_offset[r].
constexpr int kScanGroup = 1024, kScanBlock = 256, kScanPerThread = kScanGroup / kScanBlock;
__global__ __launch_bounds__(kScanBlock) void mesh_scan_groups_kernel(const unsigned* __restrict__ row_count, int n_rows,
                                                                       unsigned* __restrict__ row_offset,
                                                                       unsigned* __restrict__ group_sum) {
    __shared__ unsigned s_wave[kScanBlock / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int first = blockIdx.x * kScanGroup + (int)threadIdx.x * kScanPerThread;
    unsigned v[kScanPerThread], mine = 0u;
#pragma unroll
    for (int q = 0; q < kScanPerThread; ++q) {
        v[q] = (first + q < n_rows) ? row_count[first + q] : 0u;
        mine += v[q];
    }
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    unsigned before = 0u;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    unsigned run = before + inc - mine;
#pragma unroll
    for (int q = 0; q < kScanPerThread; ++q) {
        if (first + q < n_rows) row_offset[first + q] = run;
        run += v[q];
    }
    if (threadIdx.x == kScanBlock - 1) group_sum[blockIdx.x] = run;       // a group holds < 2^32 triangles (1024 * 2046 * 5)
}

constexpr int kTopBlock = 1024;
__global__ __launch_bounds__(kTopBlock) void mesh_scan_top_kernel(const unsigned* __restrict__ group_sum, int n_groups,
                                                                   unsigned long long* __restrict__ group_base,
                                                                   unsigned long long* __restrict__ total) {
    __shared__ unsigned long long s_wave[kTopBlock / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0ull;
    __syncthreads();
    for (int tile = 0; tile < n_groups; tile += kTopBlock) {
        const int g = tile + (int)threadIdx.x;
        const unsigned long long mine = g < n_groups ? (unsigned long long)group_sum[g] : 0ull;
        unsigned long long inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        unsigned long long before = s_base;
        for (int w = 0; w < wv; ++w) before += s_wave[w];
        if (g < n_groups) group_base[g] = before + inc - mine;
        __syncthreads();
        if (threadIdx.x == kTopBlock - 1) s_base = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_base;
}

// SDF::interpolate_color (sdf.cpp:164-217) at a world point
__device__ __forceinline__ float4 mesh_color(const MeshParams& p, const float4* __restrict__ crgb, double gx, double gy,
                                             double gz, unsigned& viol) {
    const double vx = (gx - p.g.origin[0]) * (double)p.g.m_div_w - 0.5;     // sdf.h:143-147
    const double vy = (gy - p.g.origin[1]) * (double)p.g.m_div_h - 0.5;
    const double vz = (gz - p.g.origin[2]) * (double)p.g.m_div_d - 0.5;
    const float fi = (float)vx, fj = (float)vy, fk = (float)vz;
    const int bi = mesh_trunc_x86(fi), bj = mesh_trunc_x86(fj), bk = mesh_trunc_x86(fk);
    const int m = p.g.m;
    // the eight independent loads first, then the reference's accumulation order
    float4 c[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ci = bi + (q >> 2), cj = bj + ((q >> 1) & 1), ck = bk + (q & 1);
        bool ok = (ci >= 0) & (cj >= 0) & (ck >= 0) & (ci < m) & (cj < m) & (ck < m);
        if (ok && (ci < p.g.xs || ci >= p.g.xe)) { viol = 1u; ok = false; }
        c[q] = ok ? crgb[((long long)(ci - p.g.xs) * m + cj) * m + ck] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // {Color_W, R, G, B}
    }
    float w_sum = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ci = bi + (q >> 2), cj = bj + ((q >> 1) & 1), ck = bk + (q & 1);
        const float volume = (fabsf((float)ci - fi) + fabsf((float)cj - fj)) + fabsf((float)ck - fk);
        if (c[q].x > 0.0f) {
            if ((double)volume < 0.00001) return make_float4(c[q].y, c[q].z, c[q].w, 1.0f);   // stored values, not / 255
            const float w = 1.0f / volume;
            w_sum += w;
            r += w * c[q].y;
            g += w * c[q].z;
            b += w * c[q].w;
        }
    }
    const float aux = (float)((double)w_sum * 255.0);
    return make_float4(r / aux, g / aux, b / aux, 1.0f);
}

// Triangle descriptors: one wavefront per row (as in the count kernel); rows without triangles return at once.
// desc = row << 23 | k << 11 | case << 3 | t  (k < 2^12 and row < 2^24 for m <= 4096, tsdf_create's limit)
__global__ __launch_bounds__(kMeshBlock) void mesh_list_kernel(MeshParams p, const float2* __restrict__ dw,
                                                                const unsigned* __restrict__ row_count,
                                                                const unsigned* __restrict__ row_offset,
                                                                const unsigned long long* __restrict__ group_base,
                                                                unsigned long long* __restrict__ desc,
                                                                unsigned long long capacity) {
    int i, j0;
    if (!mesh_block_rows(p, i, j0)) return;
    const int inner = p.g.m - 2;
    // most workgroups own four rows without a triangle: leave before the table load and its barrier
    {
        const int r0 = (i - p.ci0) * inner + (j0 - 1);
        unsigned any = 0u;
#pragma unroll
        for (int w = 0; w < kMeshRowsPerBlock; ++w) any |= (j0 + w <= inner) ? row_count[r0 + w] : 0u;
        if (any == 0u) return;                               // workgroup-uniform
    }
    __shared__ unsigned char s_ntri[256];
    s_ntri[threadIdx.x] = kMcNumTri[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = j0 + (int)(threadIdx.x >> 6);
    if (j > inner) return;                                   // wave-uniform
    const int row = (i - p.ci0) * inner + (j - 1);
    if (row_count[row] == 0u) return;
    const MeshRowPtrs q = mesh_rows_of(p, dw, i, j);
    unsigned long long tri_base = group_base[row >> 10] + row_offset[row];
    for (int kb = 0; kb <= inner; kb += kMeshStep) {
        int c0, c1;
        mesh_two_cubes(p, q, kb, lane, c0, c1);
        const unsigned n0 = s_ntri[c0], n1 = s_ntri[c1];
        unsigned inc = n0 + n1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        const unsigned n_step = __shfl(inc, 63, 64);
        if (n_step == 0u) continue;                          // wave-uniform
        unsigned long long at = tri_base + (inc - n0 - n1);
        const unsigned long long rk0 = ((unsigned long long)row << 23) | ((unsigned long long)(kb + 2 * lane) << 11);
        for (unsigned t = 0; t < n0; ++t, ++at)
            if (at < capacity) desc[at] = rk0 | ((unsigned long long)c0 << 3) | t;
        const unsigned long long rk1 = rk0 + (1ull << 11);
        for (unsigned t = 0; t < n1; ++t, ++at)
            if (at < capacity) desc[at] = rk1 | ((unsigned long long)c1 << 3) | t;
        tri_base += n_step;
    }
}

// One thread per output vertex.
__global__ __launch_bounds__(kMeshBlock) void mesh_vertex_kernel(MeshParams p, const float2* __restrict__ dw,
                                                                  const float4* __restrict__ crgb,
                                                                  const unsigned long long* __restrict__ desc,
                                                                  unsigned long long n_vertices,
                                                                  float* __restrict__ verts, float4* __restrict__ colors,
                                                                  unsigned* __restrict__ violations) {
    __shared__ signed char s_tri[256][16];
    reinterpret_cast<int4*>(&s_tri[0][0])[threadIdx.x] = reinterpret_cast<const int4*>(&kMcTri[0][0])[threadIdx.x];
    __syncthreads();
    const unsigned long long e = (unsigned long long)blockIdx.x * kMeshBlock + threadIdx.x;
    if (e >= n_vertices) return;
    const unsigned long long tri = e / 3ull;
    const int v = (int)(e - tri * 3ull);
    const unsigned long long ds = desc[tri];
    const int row = (int)(ds >> 23), ck = (int)((ds >> 11) & 4095ull), cidx = (int)((ds >> 3) & 255ull), t = (int)(ds & 7ull);
    const int inner = p.g.m - 2;
    const int i = p.ci0 + row / inner, j = 1 + row % inner;
    const int edge = s_tri[cidx][3 * t + v];
    // edge joins corners ea -> eb (marching_cubes_sdf.cpp:146-169)
    const int ea = (edge < 8) ? edge : edge - 8;
    const int eb = (edge < 8) ? ((edge & 4) | ((edge + 1) & 3)) : edge - 4;
    // corner c: x high for c in {1,2,5,6}, y high for c >= 4, z high for c in {2,3,6,7}   (:129-141)
    const int ha[3] = {(ea & 1) ^ ((ea >> 1) & 1), (ea >> 2) & 1, (ea >> 1) & 1};
    const int hb[3] = {(eb & 1) ^ ((eb >> 1) & 1), (eb >> 2) & 1, (eb >> 1) & 1};
    const long long m = p.g.m, zy = m * m;
    const long long g0 = ((long long)(i - p.g.xs) * m + j) * m + ck;
    const float va = dw[g0 + ha[0] * zy + ha[1] * m + ha[2]].x;
    const float vb = dw[g0 + hb[0] * zy + hb[1] * m + hb[2]].x;
    // corner positions (marching_cubes_sdf.cpp:121-141), float; min_p = 0 (setBBox, :55-65)
    const float min_p = 0.0f, fm = (float)p.g.m;
    const int idx3[3] = {i, j, ck};
    float pa[3], pb[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = min_p + (p.extent[a] - min_p) * (float)idx3[a] / fm;
        const float hi = lo + (p.extent[a] - min_p) / fm;
        pa[a] = ha[a] ? hi : lo;
        pb[a] = hb[a] ? hi : lo;
    }
    const float mu = (p.iso - va) / (vb - va);                                 // :87-94
    float o3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) o3[a] = pa[a] + mu * (pb[a] - pa[a]);
    float* dst = &verts[e * 3ull];
    dst[0] = o3[0]; dst[1] = o3[1]; dst[2] = o3[2];
    if (colors) {                                                              // sdf.cpp:353-383
        unsigned viol = 0u;
        colors[e] = mesh_color(p, crgb, (double)o3[0] + p.g.origin[0], (double)o3[1] + p.g.origin[1],
                               (double)o3[2] + p.g.origin[2], viol);
        if (viol) atomicOr(violations, 1u);
    }
}

hipError_t launch_mesh_count(hipStream_t s, const MeshParams& p, const float2* dw, unsigned* row_count,
                             unsigned* row_offset, unsigned* group_sum, unsigned long long* group_base,
                             unsigned long long* total) {
    const long long n_rows = mesh_rows(p);
    if (n_rows <= 0) return hipMemsetAsync(total, 0, sizeof(unsigned long long), s);
    hipError_t e = hipMemsetAsync(row_count, 0, (size_t)n_rows * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    mesh_count_kernel<<<dim3(mesh_count_grid(p)), dim3(kMeshBlock), 0, s>>>(p, dw, row_count);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n_groups = (int)mesh_scan_groups(n_rows);
    mesh_scan_groups_kernel<<<dim3(n_groups), dim3(kScanBlock), 0, s>>>(row_count, (int)n_rows, row_offset, group_sum);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    mesh_scan_top_kernel<<<dim3(1), dim3(kTopBlock), 0, s>>>(group_sum, n_groups, group_base, total);
    return hipGetLastError();
}

hipError_t launch_mesh_emit(hipStream_t s, const MeshParams& p, const float2* dw, const float4* crgb,
                            const unsigned* row_count, const unsigned* row_offset, const unsigned long long* group_base,
                            unsigned long long* desc, float* verts, float4* colors, unsigned long long n_triangles,
                            unsigned* violations) {
    if (mesh_rows(p) <= 0 || n_triangles == 0) return hipSuccess;
    mesh_list_kernel<<<dim3(mesh_grid(p)), dim3(kMeshBlock), 0, s>>>(p, dw, row_count, row_offset, group_base, desc, n_triangles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const unsigned long long n_vertices = 3ull * n_triangles;
    const unsigned long long blocks = (n_vertices + kMeshBlock - 1) / kMeshBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    mesh_vertex_kernel<<<dim3((unsigned)blocks), dim3(kMeshBlock), 0, s>>>(p, dw, crgb, desc, n_vertices, verts, colors, violations);
    return hipGetLastError();
}

}  // namespace tsdf
