// tsdf_device.h -- parameter blocks and launch entry points shared by the HIP kernels (integrate_kernels.hip,
// track_kernels.hip, volume_kernels.hip, preproc_kernels.hip, mesh_kernels.hip) and the C-ABI host side (api_*.cpp).  gfx950 only.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace tsdf {

// Width of one reduction row produced by the tracker kernels (doubles).
//   [0..20]  upper triangle of sum J J^T (row-major: 00 01 .. 05 11 12 .. 55)
//   [21..26] sum r J
//   [27]     number of terms added (stale re-adds included)
//   [28]     look-ups that left this rank's slab+halo (must be 0)
//   [29]     owned samples with all 13 look-ups valid
//   [30]     owned in-grid samples
//   [31]     out-of-grid samples   [32] NaN samples   [33] sampled pixels
constexpr int kRedWidth = 34;
constexpr int kRedAllreduce = 30;   // the leading part that is summed over ranks
// partial-row layout of track_kernel (workgroup rows and shard rows): [5*q + d] (q = 0..5, d = 0..4) = J[q]*J[(q+d)%6] for
// d <= 3 and r*J[q] for d = 4, then counters
constexpr int kPartTerms = 30, kPartViol = 31, kPartOk = 32, kPartInOwned = 33, kPartOog = 34, kPartNan = 35,
              kPartSamples = 36, kPartWidth = 40;
constexpr int kTrackBlock = 384;                 // threads per tracker workgroup (640x480: 714 workgroups = 2.8 per CU; 512 threads lost a resident workgroup per CU)
constexpr int kIntegrateBlock = 256;             // threads per integrate workgroup
constexpr int kTrackShards = 8;                  // fan-in shards of the in-launch fold (blockIdx % 8: one per XCD)
constexpr int kShardSlotDoubles = 80;            // pinned host slot of a shard row: 40 {value, pass word} pairs of 16 bytes -- a value
                                                 // and the word that validates it arrive in ONE store, so no fence separates them

// What travels next to a shard-row value in its 16-byte {value, word} pair (host side of the tracker fan-in): the pass
// word mixed with the value's own bits.  The host accepts a value when shard_pair_word(value bits, pass) equals the word
// it reads next to it, so a pair torn anywhere between the store instruction and host memory cannot pass a stale value.
#ifdef __HIPCC__
__host__ __device__
#endif
inline unsigned long long shard_pair_word(unsigned long long value_bits, unsigned long long pass_word) {
    return pass_word ^ (value_bits * 0x9E3779B97F4A7C15ull);
}

// partial row (kPartWidth) -> result row (kRedWidth): the mapping track_kernel's last workgroup applies, for the host
// side of the fan-in
inline void track_unpack_row(const double* tot, double* red) {
    int e = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            const int d = b - a;
            red[e++] = (d <= 3) ? tot[5 * a + d] : tot[5 * b + (6 - d)];
        }
    for (int a = 0; a < 6; ++a) red[21 + a] = tot[5 * a + 4];
    red[27] = tot[kPartTerms]; red[28] = tot[kPartViol]; red[29] = tot[kPartOk]; red[30] = tot[kPartInOwned];
    red[31] = tot[kPartOog]; red[32] = tot[kPartNan]; red[33] = tot[kPartSamples];
}

// Geometry of the stored part of the volume.  Device layout: one float2 {D,W} per voxel
// (and one float4 {Color_W,R,G,B} when colour is kept), reference index order
// idx = m*m*i + m*j + k restricted to i in [xs, xe)  (x-slab + halo), k fastest.
struct Grid {
    int32_t m;
    int32_t xs, xe;          // stored x layers [xs, xe)  (block-cyclic: those of block 0, clipped to the grid)
    int32_t own_x0, own_x1;  // owned x layers (slab without halo; block-cyclic: those of block 0)
    float cell_w, cell_h, cell_d;     // extent / (float)m, float quotients (sdf.h:154-156)
    float m_div_w, m_div_h, m_div_d;  // m / extent, float quotients (sdf.cpp:19-21)
    double origin[3];
    float delta, epsilon;
    // Block-cyclic placement (blk_own > 0; tsdf_config::slab_stride): the handle owns the layers
    // [own_x0 + b * blk_stride, own_x0 + b * blk_stride + blk_own) of blocks b = 0 .. n_blocks - 1 and stores each block with
    // its halo: blk_layers = blk_own + 2 * halo layers per block, back to back in memory, block b's first stored layer
    // being the global layer blk_first + b * blk_stride (blk_first = own_x0 - halo; layers outside [0, m) -- the halo of
    // a block at the edge of the grid -- exist in memory and are never listed, read or written).  The stored ranges of
    // two blocks never overlap (blk_stride >= blk_layers).  blk_magic = 2^32 / blk_layers + 1: local layer / blk_layers
    // as a multiply-high (exact for local layers < 2^32 / blk_layers^2: checked at creation).
    int32_t blk_own, blk_layers, blk_stride, blk_first, n_blocks;
    uint32_t blk_magic;
};

#if defined(__HIPCC__)
#define TSDF_HD __host__ __device__
#else
#define TSDF_HD
#endif
// stored x layers of the handle (every block with its halo)
TSDF_HD inline int grid_stored_layers(const Grid& g) { return g.blk_own > 0 ? g.n_blocks * g.blk_layers : g.xe - g.xs; }
// local stored layer -> global x layer (outside [0, m) for the padding layers of a block at the edge)
TSDF_HD inline int grid_global_layer(const Grid& g, int il) {
    if (g.blk_own <= 0) return il + g.xs;
    const int b = (int)(((unsigned long long)(unsigned)il * g.blk_magic) >> 32);
    return g.blk_first + b * g.blk_stride + (il - b * g.blk_layers);
}
// is the global x layer one of the handle's OWN layers?
TSDF_HD inline bool grid_owns_layer(const Grid& g, int gi) {
    if (g.blk_own <= 0) return gi >= g.own_x0 && gi < g.own_x1;
    if (gi < g.own_x0 || gi >= g.m) return false;
    return (gi - g.own_x0) % g.blk_stride < g.blk_own;
}
// The block whose stored layers hold (or would hold) the global layer gi: its index, clamped to the handle's blocks.
TSDF_HD inline int grid_block_of(const Grid& g, int gi) {
    if (g.blk_own <= 0) return 0;
    int t = gi - g.blk_first;
    int b = t < 0 ? 0 : t / g.blk_stride;
    return b >= g.n_blocks ? g.n_blocks - 1 : b;
}

struct IntegrateParams {
    Grid g;
    double rot_inv[9];
    double rot_inv_trans[3];
    double K[9];
    int32_t width, height;
    int32_t pix_su, pix_sv;  // record index of pixel (col,row) = col*pix_su + row*pix_sv
    int32_t with_color;
};

struct TrackParams {
    Grid g;
    double rot[9];
    double trans[3];
    double rpm[54];          // r1p r1m r2p r2m r3p r3m  (camera_tracking.cpp:92-145)
    float v_h;
    float vh2[3];            // 2 v_h / m_div_{w,h,d}   (camera_tracking.cpp:13-17)
    float wh2;               // 2 * w_h as a float product (camera_tracking.cpp:331)
    int32_t n_samples;
    int32_t stale_carry;
    int32_t carry_threads;   // >= 1: OpenMP threads of the reference run whose carry resets are reproduced
    int32_t ncols, nrows;    // sample grid: sample n = col * nrows + row (columns outer, camera_tracking.cpp:162-163)
    // non-null: the frame's packing is deferred (tsdf_set_frame_device) and the samples are read from the caller's
    // xyz plane: sample (col, row) = pixel (col * pixel_stride, row * pixel_stride) of a plane_width-wide image
    const float* xyz_plane = nullptr;
    int32_t plane_width = 0, pixel_stride = 1;
    float4* sample_list_out = nullptr;   // with xyz_plane: every workgroup also writes its own samples here (for the later passes)
};

// What pack_kernel needs: the frame's planes (device memory; nrm / rgb may be null) and where the packed records go.
struct PackArgs {
    const float* xyz = nullptr; const float* nrm = nullptr; const uint8_t* rgb = nullptr;
    int32_t width = 0, height = 0, stride = 1;
    int32_t pix_su = 0, pix_sv = 0;       // record index of pixel (col,row) = col*pix_su + row*pix_sv
    float4* pn = nullptr;                 // kPixelRecordBytes per pixel
    float4* samples = nullptr;            // ncols x nrows tracker samples (null: not written)
    int32_t ncols = 0, nrows = 0;
    int32_t color_layout = 0;             // 0: 24-byte {P,N} records; 1: 32-byte records (volume with colour)
};

// mesh extraction (mesh_kernels.hip): cubes with base voxel layer i in [ci0, ci1), j,k in [1, m-2]
struct MeshParams {
    Grid g;
    float extent[3];         // width, height, depth (setBBox, marching_cubes_sdf.cpp:55-65)
    float iso;
    int32_t ci0, ci1;
};

// cumulative device counters (unsigned long long each)
enum Counter { kCntUpdatedOwned = 0, kCntUpdatedHalo = 1, kCntItems = 2, kCntOverflowItems = 3, kNumCounters = 4 };

// pinned host word + the value a launch stores there once the caller's device planes have been read (tsdf_device_frame_released)
struct ReleaseWord {
    unsigned long long* word = nullptr; unsigned long long ticket = 0;
    unsigned long long* items_word = nullptr;   // pinned host word (may be null): integrate_kernel leaves the launch's work-item count there
};
hipError_t launch_release(hipStream_t s, const ReleaseWord& rel);   // behind a launch_pack of borrowed device planes

hipError_t launch_fill(hipStream_t s, const Grid& g, float2* dw, float4* crgb, float d0);
hipError_t launch_pack(hipStream_t s, const PackArgs& a);
// worklist: integrate_worklist_bytes(g) bytes, zero before the first launch; work_count:
// integrate_bookkeeping_words() unsigned, zero before the first launch; n_blocks: persistent grid size (CUs x
// integrate_blocks_per_cu(), a multiple of 8).
size_t integrate_worklist_entries(const Grid& g);
size_t integrate_band_region_entries(const Grid& g);   // entries of the band regions in front of the overflow region
size_t integrate_worklist_bytes(const Grid& g);      // 32-byte item descriptors: band regions + overflow region
constexpr size_t kPixelRecordBytes = 32;             // per pixel: two float4 records (24 of them used when the volume has no colour)
size_t integrate_bookkeeping_words();
int integrate_blocks_per_cu();
hipError_t launch_integrate(hipStream_t s, const IntegrateParams& p, float2* dw, float4* crgb,
                            const float4* pn, unsigned long long* counters,
                            void* worklist, unsigned* work_count, int n_blocks,
                            unsigned launch_parity, unsigned long long* wg_counts /* 2 per workgroup, zero at start */,
                            const PackArgs* pack = nullptr /* the frame's records are still to be packed: done inside this launch */,
                            const ReleaseWord* release = nullptr /* with pack of borrowed planes: told to the host once they are read */);
// the two halves of launch_integrate, for a caller that can issue the list before the frame's records are complete
hipError_t launch_integrate_list(hipStream_t s, const IntegrateParams& p, void* worklist, unsigned* work_count,
                                 unsigned launch_parity, const PackArgs* pack = nullptr);
hipError_t launch_integrate_items(hipStream_t s, const IntegrateParams& p, float2* dw, float4* crgb,
                                  const float4* pn, unsigned long long* counters,
                                  void* worklist, unsigned* work_count, int n_blocks,
                                  unsigned launch_parity, unsigned long long* wg_counts, const ReleaseWord* release = nullptr);
// One tracker pass = one launch.  partials: track_partials_doubles(n_samples) doubles (per-workgroup rows + shard rows);
// ctr: track_fold_counter_words() unsigned, zero before the first pass; red_dev (may be null): kRedWidth doubles for an
// in-stream all-reduce; host_row (pinned or a registered shared segment, may be null): kRedWidth doubles + one 64-bit
// word that receives `word` after the row is complete; host_shards (pinned, may be null): when given, the shard rows go
// to the host instead (kShardSlotDoubles per shard, word at [kPartWidth]) and the host adds them; `pass` tags the rows.
// Device-side exchange of the result rows between the ranks of one node (tsdf_comm_init_peer): every rank owns a
// buffer of n x 2 slots of kPeerSlotBytes (writer rank, pass parity) in uncached device memory, mapped into every other
// rank through a HIP IPC handle.  The workgroup that finishes a rank's row stores it into its slot of EVERY rank's
// buffer (its own included), releases the pass word behind it at system scope, waits for the n words of its own buffer
// and adds the leading kRedAllreduce entries in rank order -- the same order and bits on every rank.
constexpr size_t kPeerSlotBytes = 512;            // 34 doubles + the word, padded
constexpr int kPeerMaxRanks = 64;
struct PeerExchange {
    char* const* bases = nullptr;   // device array of n pointers: rank r's buffer as mapped into this process
    int n = 0, rank = 0;            // n == 0: no exchange
    unsigned parity = 0;            // which slot of the pair this pass uses
    unsigned long long word = 0;    // what is released behind a row, and waited for
    long long timeout_ticks = 0;    // wall_clock64() ticks (100 MHz) a rank waits for the others before it gives up
};
class AqlQueue;   // aql_queue.hpp
hipError_t launch_track_folded(hipStream_t s, const TrackParams& p, const float2* dw, const float4* samples,
                               double* partials, unsigned* ctr, double* red_dev, double* host_row, double* host_shards,
                               unsigned long long word, unsigned long long pass, const PeerExchange* peers = nullptr,
                               AqlQueue* aql = nullptr /* dispatch through the library's own queue instead of the stream (falls
                                                          back to the stream when the queue refuses) */,
                               bool* went_through_queue = nullptr /* out: the queue took the dispatch */);
// what AqlQueue::init needs to know about track_kernel: the size of its explicit arguments and how its symbol begins
size_t track_kernel_explicit_arg_bytes();
const char* track_kernel_symbol_prefix();
const char* track_kernel_build_id();       // TSDF_BUILD_ID of this build: the stand-alone code object must carry the same
// the same exchange for a row that is already in red_dev (tsdf_allreduce): one wavefront; n_sum leading entries are added
hipError_t launch_peer_exchange(hipStream_t s, const PeerExchange& px, double* red_dev, int n_sum, double* host_row,
                                unsigned long long host_word);
// NaN bit patterns a row's term count (entry 27) carries when the hand-off failed
constexpr unsigned long long kRowPoisonStale = 0x7ff8000000000000ull;
constexpr unsigned long long kRowPoisonPeerTimeout = 0x7ff8000000000001ull;
int track_num_shards(int32_t n_samples);
size_t track_fold_counter_words();
hipError_t launch_track_publish(hipStream_t s, const double* red_dev, double* red_host, unsigned long long seq);
int track_num_blocks(int32_t n_samples);
size_t track_partials_doubles(int32_t n_samples);
// depth -> z plane; with `minmax` (2 words) also min bits / ~max bits of the valid depths (0xffffffff = none)
hipError_t launch_depth_to_z(hipStream_t s, const uint16_t* d16, const float* dflt, float scale, int n, float* z,
                             unsigned* minmax);
struct BilateralGrid { int gx, gy, gz; float zmin; int cut; float zcut; };   // cells per axis (padding included), depth of cell 2;
                                                                              // cut: pixels at or beyond zcut are dropped first
// false: sigma_s outside [1, 30], or a depth range / sigma_r that is not a sane cell count
bool bilateral_grid_plan(int w, int h, float sigma_s, float sigma_r, float zmin, float zmax, BilateralGrid* g);
// bg = nullptr: windowed bilateral of radius R; else the bilateral grid in grid_a / grid_b (gx*gy*gz cells each)
hipError_t launch_preproc(hipStream_t s, int w, int h, const float* K /*fx fy cx cy*/, int R, float sigma_s, float sigma_r,
                          int nr, float max_change, const BilateralGrid* bg, float2* grid_a, float2* grid_b,
                          const float* z, float* zf, float* xyz, float* nrm);
// row_count / row_offset: mesh_rows(p) entries; group_sum / group_base: mesh_scan_groups(rows) entries (groups of
// 1024 rows); total: one 64-bit word (triangles).  Row r starts at triangle group_base[r >> 10] + row_offset[r].
inline long long mesh_rows(const MeshParams& p) { return (long long)(p.ci1 > p.ci0 ? p.ci1 - p.ci0 : 0) * (p.g.m - 2); }
inline long long mesh_scan_groups(long long n_rows) { return (n_rows + 1023) >> 10; }
hipError_t launch_mesh_count(hipStream_t s, const MeshParams& p, const float2* dw, unsigned* row_count,
                             unsigned* row_offset, unsigned* group_sum, unsigned long long* group_base,
                             unsigned long long* total);
hipError_t launch_mesh_emit(hipStream_t s, const MeshParams& p, const float2* dw, const float4* crgb,
                            const unsigned* row_count, const unsigned* row_offset, const unsigned long long* group_base,
                            unsigned long long* desc /* one per triangle */, float* verts, float4* colors,
                            unsigned long long n_triangles, unsigned* violations);
hipError_t launch_sample(hipStream_t s, const Grid& g, const float2* dw, const double* vox, int32_t n,
                         float* val, int32_t* ok);
hipError_t launch_split(hipStream_t s, const float2* dw, float* d, float* w, int64_t n);
hipError_t launch_merge(hipStream_t s, float2* dw, const float* d, const float* w, int64_t n);
hipError_t launch_split4(hipStream_t s, const float4* c, float* a, float* r, float* g, float* b, int64_t n);
hipError_t launch_merge4(hipStream_t s, float4* c, const float* a, const float* r, const float* g, const float* b, int64_t n);

}  // namespace tsdf
