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
// The triangle table is mc_tables.h (tools/gen_mc_tables.py): the reference table's polygons in all 256
// cases, own diagonals -- same vertices, same triangle count per cube.
//
// Three launches, all HBM-bound on the 8 B/voxel D,W sweep:
//   mesh_count_kernel   one workgroup per (i,j) row of cubes: case numbers, triangles per row
//   mesh_scan_kernel    exclusive scan of the row counts (one workgroup; rows are few: (m-2)^2 per layer)
//   mesh_emit_kernel    rows with triangles only: case numbers again, in-row scan, vertices (+ colours)
// Recomputing the case numbers in the emit pass costs a second sweep of the rows that have surface
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

// the eight corners of cube (i,j,k): returns the case number (0 when the W gate fails) and the corner values
__device__ __forceinline__ int mesh_cube(const MeshParams& p, const float2* __restrict__ dw, int i, int j, int k,
                                         float leaf[8]) {
    const long long m = p.g.m;
    const long long g0 = ((long long)(i - p.g.xs) * m + j) * m + k;
    const long long zy = m * m;
    const long long off[8] = {0, zy, zy + 1, 1, m, m + zy, m + zy + 1, m + 1};
    bool all = true;
    int cubeindex = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float2 v = dw[g0 + off[c]];
        leaf[c] = v.x;
        all = all && (v.y > 0.0f);
        if (v.x < p.iso) cubeindex |= 1 << c;
    }
    return all ? cubeindex : 0;
}

// rows: r = (i - ci0) * (m-2) + (j-1).  Workgroups are dealt round-robin to the 8 XCDs; give every XCD one
// contiguous eighth of the rows so that the rows (i,j+1), (i+1,j), (i+1,j+1) a row shares with its neighbours
// are found in that XCD's L2.
__device__ __forceinline__ int mesh_row_of_block(unsigned block, int n_rows) {
    const int per = (n_rows + 7) >> 3;
    return (int)(block & 7u) * per + (int)(block >> 3);
}
__host__ inline unsigned mesh_grid(long long n_rows) { return (unsigned)(((n_rows + 7) >> 3) << 3); }

__global__ __launch_bounds__(kMeshBlock) void mesh_count_kernel(MeshParams p, const float2* __restrict__ dw,
                                                                 unsigned* __restrict__ row_count) {
    __shared__ unsigned s_sum[kMeshBlock / 64];
    const int inner = p.g.m - 2;
    const int n_rows = (p.ci1 - p.ci0) * inner;
    const int row = mesh_row_of_block(blockIdx.x, n_rows);
    if (row >= n_rows) return;
    const int i = p.ci0 + row / inner, j = 1 + row % inner;
    unsigned n = 0;
    for (int k = 1 + (int)threadIdx.x; k <= inner; k += kMeshBlock) {
        float leaf[8];
        n += kMcNumTri[mesh_cube(p, dw, i, j, k, leaf)];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < kMeshBlock / 64; ++w) t += s_sum[w];
        row_count[row] = t;
    }
}

// exclusive scan of row_count -> row_offset (triangles), total -> *total.  One workgroup of 1024 threads walks
// the rows in tiles of 4096.
constexpr int kScanBlock = 1024, kScanPerThread = 4;
__global__ __launch_bounds__(kScanBlock) void mesh_scan_kernel(const unsigned* __restrict__ row_count, int n_rows,
                                                                unsigned long long* __restrict__ row_offset,
                                                                unsigned long long* __restrict__ total) {
    __shared__ unsigned long long s_wave[kScanBlock / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0ull;
    __syncthreads();
    for (int tile = 0; tile < n_rows; tile += kScanBlock * kScanPerThread) {
        const int first = tile + (int)threadIdx.x * kScanPerThread;
        unsigned v[kScanPerThread];
        unsigned long long mine = 0ull;
#pragma unroll
        for (int q = 0; q < kScanPerThread; ++q) {
            v[q] = (first + q < n_rows) ? row_count[first + q] : 0u;
            mine += v[q];
        }
        // inclusive scan over the wave, then over the 16 waves
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
        unsigned long long run = before + inc - mine;
#pragma unroll
        for (int q = 0; q < kScanPerThread; ++q) {
            if (first + q < n_rows) row_offset[first + q] = run;
            run += v[q];
        }
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_base = run;
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
    float w_sum = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
    const int m = p.g.m;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ci = bi + (q >> 2), cj = bj + ((q >> 1) & 1), ck = bk + (q & 1);
        const float volume = (fabsf((float)ci - fi) + fabsf((float)cj - fj)) + fabsf((float)ck - fk);
        bool ok = (ci >= 0) & (cj >= 0) & (ck >= 0) & (ci < m) & (cj < m) & (ck < m);
        if (ok && (ci < p.g.xs || ci >= p.g.xe)) { viol = 1u; ok = false; }
        if (!ok) continue;
        const float4 c = crgb[((long long)(ci - p.g.xs) * m + cj) * m + ck];       // {Color_W, R, G, B}
        if (c.x > 0.0f) {
            if ((double)volume < 0.00001) return make_float4(c.y, c.z, c.w, 1.0f);   // stored values, not / 255
            const float w = 1.0f / volume;
            w_sum += w;
            r += w * c.y;
            g += w * c.z;
            b += w * c.w;
        }
    }
    const float aux = (float)((double)w_sum * 255.0);
    return make_float4(r / aux, g / aux, b / aux, 1.0f);
}

__global__ __launch_bounds__(kMeshBlock) void mesh_emit_kernel(MeshParams p, const float2* __restrict__ dw,
                                                                const float4* __restrict__ crgb,
                                                                const unsigned* __restrict__ row_count,
                                                                const unsigned long long* __restrict__ row_offset,
                                                                float* __restrict__ verts, float4* __restrict__ colors,
                                                                unsigned long long capacity,
                                                                unsigned* __restrict__ violations) {
    const int inner = p.g.m - 2;
    const int n_rows = (p.ci1 - p.ci0) * inner;
    const int row = mesh_row_of_block(blockIdx.x, n_rows);
    if (row >= n_rows || row_count[row] == 0u) return;      // workgroup-uniform
    __shared__ signed char s_tri[256][16];
    __shared__ unsigned s_wave[kMeshBlock / 64];
    __shared__ unsigned s_base;
    {
        const int4* src = reinterpret_cast<const int4*>(&kMcTri[0][0]);
        reinterpret_cast<int4*>(&s_tri[0][0])[threadIdx.x] = src[threadIdx.x];      // 256 x 16 bytes
    }
    if (threadIdx.x == 0) s_base = 0u;
    __syncthreads();
    const int i = p.ci0 + row / inner, j = 1 + row % inner;
    const unsigned long long row_first = row_offset[row];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float min_p = 0.0f;                                // setBBox, marching_cubes_sdf.cpp:55-65
    const float ext[3] = {p.extent[0], p.extent[1], p.extent[2]};
    const float fm = (float)p.g.m;
    unsigned viol = 0u;

    for (int k0 = 1; k0 <= inner; k0 += kMeshBlock) {
        const int k = k0 + (int)threadIdx.x;
        float leaf[8];
        int cubeindex = 0;
        if (k <= inner) cubeindex = mesh_cube(p, dw, i, j, k, leaf);
        const unsigned nt = kMcNumTri[cubeindex];
        // exclusive scan of nt over the workgroup, k order
        unsigned inc = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        unsigned before = s_base;
        for (int w = 0; w < wv; ++w) before += s_wave[w];
        const unsigned first = before + inc - nt;

        if (nt) {
            // corner positions (marching_cubes_sdf.cpp:121-141), float
            const int idx3[3] = {i, j, k};
            float lo[3], hi[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = min_p + (ext[a] - min_p) * (float)idx3[a] / fm;
                hi[a] = lo[a] + (ext[a] - min_p) / fm;
            }
            // corner c: x high for c in {1,2,5,6}, y high for c >= 4, z high for c in {2,3,6,7}
            for (unsigned t = 0; t < nt; ++t) {
                const unsigned long long tri = row_first + first + t;
                if (tri >= capacity) break;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int e = s_tri[cubeindex][3 * t + v];
                    // edge e joins corners ea, eb (marching_cubes_sdf.cpp:146-169)
                    const int ea = (e < 8) ? e : e - 8;
                    const int eb = (e < 8) ? ((e & 4) | ((e + 1) & 3)) : e - 4;
                    float pa[3], pb[3];
                    pa[0] = (((ea & 1) ^ ((ea >> 1) & 1)) ? hi[0] : lo[0]);
                    pa[1] = ((ea & 4) ? hi[1] : lo[1]);
                    pa[2] = ((ea & 2) ? hi[2] : lo[2]);
                    pb[0] = (((eb & 1) ^ ((eb >> 1) & 1)) ? hi[0] : lo[0]);
                    pb[1] = ((eb & 4) ? hi[1] : lo[1]);
                    pb[2] = ((eb & 2) ? hi[2] : lo[2]);
                    float va = 0.0f, vb = 0.0f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {            // register array indexed by a runtime value: select
                        va = (c == ea) ? leaf[c] : va;
                        vb = (c == eb) ? leaf[c] : vb;
                    }
                    const float mu = (p.iso - va) / (vb - va);                     // :87-94
                    float o3[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) o3[a] = pa[a] + mu * (pb[a] - pa[a]);
                    float* dst = &verts[(tri * 3ull + v) * 3ull];
                    dst[0] = o3[0]; dst[1] = o3[1]; dst[2] = o3[2];
                    if (colors)                                                    // sdf.cpp:353-383
                        colors[tri * 3ull + v] = mesh_color(p, crgb, (double)o3[0] + p.g.origin[0],
                                                            (double)o3[1] + p.g.origin[1], (double)o3[2] + p.g.origin[2], viol);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == kMeshBlock - 1) s_base = first + nt;
        __syncthreads();
    }
    if (viol) atomicOr(violations, 1u);
}

hipError_t launch_mesh_count(hipStream_t s, const MeshParams& p, const float2* dw, unsigned* row_count,
                             unsigned long long* row_offset, unsigned long long* total) {
    const long long n_rows = mesh_rows(p);
    if (n_rows <= 0) return hipMemsetAsync(total, 0, sizeof(unsigned long long), s);
    mesh_count_kernel<<<dim3(mesh_grid(n_rows)), dim3(kMeshBlock), 0, s>>>(p, dw, row_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    mesh_scan_kernel<<<dim3(1), dim3(kScanBlock), 0, s>>>(row_count, (int)n_rows, row_offset, total);
    return hipGetLastError();
}

hipError_t launch_mesh_emit(hipStream_t s, const MeshParams& p, const float2* dw, const float4* crgb,
                            const unsigned* row_count, const unsigned long long* row_offset, float* verts, float4* colors,
                            unsigned long long capacity, unsigned* violations) {
    const long long n_rows = mesh_rows(p);
    if (n_rows <= 0) return hipSuccess;
    mesh_emit_kernel<<<dim3(mesh_grid(n_rows)), dim3(kMeshBlock), 0, s>>>(p, dw, crgb, row_count, row_offset, verts, colors,
                                                                          capacity, violations);
    return hipGetLastError();
}

}  // namespace tsdf
