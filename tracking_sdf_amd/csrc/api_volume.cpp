// api_volume.cpp -- the volume as the host sees it: reference-order mirrors (tsdf_download / upload), checkpoints
// (TSDFVOL2), mesh extraction (see handle.hpp).
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

// ---- mesh extraction ---------------------------------------------------------------------------------

static int refuse_cyclic(tsdf_handle* h, const char* who) {
    if (h->grid.blk_own > 0)
        return fail(h, TSDF_E_BADARG, "%s: not available for a handle with block-cyclic placement (slab_stride > 0): its stored layers are not one range", who);
    return TSDF_OK;
}

int tsdf_mesh_extract(tsdf_handle* h, float iso_level, int32_t with_color, int64_t* n_triangles) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (n_triangles) *n_triangles = 0;
    h->mesh_ntri = -1;
    if (!(iso_level >= 0.0f && iso_level < 1.0f))                 // marching_cubes_sdf.cpp:246-252
        return fail(h, TSDF_E_BADARG, "tsdf_mesh_extract: iso level %g outside [0,1)", (double)iso_level);
    if (with_color && !h->crgb) return fail(h, TSDF_E_BADARG, "tsdf_mesh_extract: the volume keeps no colour");
    const Grid& g = h->grid;
    // The cubes whose base layer the handle owns: one piece for a plain slab; for a block-cyclic handle one piece per block,
    // each a small slab of its own as far as the kernels are concerned (its stored layers start at the block's first one).
    struct Piece { MeshParams p; const float2* dw; const float4* crgb; unsigned long long n; };
    std::vector<Piece> pieces;
    const long long mm = (long long)g.m * g.m;
    try {
        const int nb = g.blk_own > 0 ? g.n_blocks : 1;
        for (int b = 0; b < nb; ++b) {
            Piece pc{};
            pc.p.g = g;
            pc.dw = h->dw; pc.crgb = h->crgb;
            if (g.blk_own > 0) {
                const int first = g.blk_first + b * g.blk_stride;
                pc.p.g.blk_own = 0;                                 // (a plain slab to the kernels)
                pc.p.g.xs = first;                                  // may be negative: only (i - xs) with i >= 0 is ever formed
                pc.p.g.xe = first + g.blk_layers > g.m ? g.m : first + g.blk_layers;
                pc.p.g.own_x0 = g.own_x0 + b * g.blk_stride; pc.p.g.own_x1 = pc.p.g.own_x0 + g.blk_own;
                pc.dw = h->dw + (long long)b * g.blk_layers * mm;
                pc.crgb = h->crgb ? h->crgb + (long long)b * g.blk_layers * mm : nullptr;
            }
            pc.p.extent[0] = h->cfg.width; pc.p.extent[1] = h->cfg.height; pc.p.extent[2] = h->cfg.depth;
            pc.p.iso = iso_level;
            pc.p.ci0 = pc.p.g.own_x0 > 1 ? pc.p.g.own_x0 : 1;
            pc.p.ci1 = pc.p.g.own_x1 < g.m - 1 ? pc.p.g.own_x1 : g.m - 1;   // cube layers [ci0, ci1): base voxels 1..m-2
            if (pc.p.ci1 > pc.p.ci0 && pc.p.g.xe < pc.p.ci1 + 1)
                return fail(h, TSDF_E_HALO, "tsdf_mesh_extract: a sharded volume needs halo >= 1 (cube layer %d reads layer %d)",
                            pc.p.ci1 - 1, pc.p.ci1);
            pieces.push_back(pc);
        }
    } catch (...) { return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: out of host memory"); }
    size_t n_rows = 0;
    for (const Piece& pc : pieces) { const size_t r = (size_t)mesh_rows(pc.p); if (r > n_rows) n_rows = r; }
    if (n_rows > (size_t)INT32_MAX) return fail(h, TSDF_E_BADARG, "tsdf_mesh_extract: too many rows");
    if (n_rows > h->mesh_rows_cap) {
        if (h->mesh_row_count) (void)hipFree(h->mesh_row_count);
        if (h->mesh_row_offset) (void)hipFree(h->mesh_row_offset);
        if (h->mesh_group_sum) (void)hipFree(h->mesh_group_sum);
        if (h->mesh_group_base) (void)hipFree(h->mesh_group_base);
        h->mesh_row_count = nullptr; h->mesh_row_offset = nullptr; h->mesh_rows_cap = 0;
        h->mesh_group_sum = nullptr; h->mesh_group_base = nullptr;
        const size_t n_groups = (size_t)mesh_scan_groups((long long)n_rows);
        if (hipMalloc((void**)&h->mesh_row_count, n_rows * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_row_offset, n_rows * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_group_sum, n_groups * sizeof(unsigned)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_group_base, n_groups * sizeof(unsigned long long)) != hipSuccess)
            return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: row tables (%zu rows)", n_rows);
        h->mesh_rows_cap = n_rows;
    }
    if (!h->mesh_total) HIP_TRY(h, hipHostMalloc((void**)&h->mesh_total, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    unsigned long long* d_total = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer((void**)&d_total, h->mesh_total, 0));
    auto count = [&](Piece& pc) -> int {
        h->mesh_total[0] = 0ull; h->mesh_total[1] = 0ull;
        HIP_TRY(h, launch_mesh_count(h->stream, pc.p, pc.dw, h->mesh_row_count, h->mesh_row_offset, h->mesh_group_sum,
                                     h->mesh_group_base, d_total));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        pc.n = h->mesh_total[0];
        return TSDF_OK;
    };
    unsigned long long n = 0;
    for (Piece& pc : pieces) {
        if ((rc = count(pc)) != TSDF_OK) return rc;
        n += pc.n;
    }
    if (n > (unsigned long long)INT64_MAX / 64) return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: %llu triangles", n);
    if (n > h->mesh_verts_cap) {
        if (h->mesh_verts) (void)hipFree(h->mesh_verts);
        if (h->mesh_desc) (void)hipFree(h->mesh_desc);
        h->mesh_verts = nullptr; h->mesh_desc = nullptr; h->mesh_verts_cap = 0;
        const size_t cap = (size_t)n + (size_t)n / 8 + 1024;          // room to grow between calls
        if (hipMalloc((void**)&h->mesh_verts, cap * 9 * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&h->mesh_desc, cap * sizeof(unsigned long long)) != hipSuccess)
            return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: %llu triangles need %zu bytes", n, cap * 9 * sizeof(float));
        h->mesh_verts_cap = cap;
    }
    if (with_color && n > h->mesh_colors_cap) {
        if (h->mesh_colors) (void)hipFree(h->mesh_colors);
        h->mesh_colors = nullptr; h->mesh_colors_cap = 0;
        const size_t cap = h->mesh_verts_cap;
        if (hipMalloc((void**)&h->mesh_colors, cap * 3 * sizeof(float4)) != hipSuccess)
            return fail(h, TSDF_E_NOMEM, "tsdf_mesh_extract: colours of %llu triangles", n);
        h->mesh_colors_cap = cap;
    }
    unsigned long long base = 0;
    for (Piece& pc : pieces) {
        if (!pc.n) continue;
        const unsigned long long n_piece = pc.n;
        if (pieces.size() > 1 && (rc = count(pc)) != TSDF_OK) return rc;       // the row tables hold the LAST piece counted: once more for this one
        h->mesh_total[1] = 0ull;
        HIP_TRY(h, launch_mesh_emit(h->stream, pc.p, pc.dw, pc.crgb, h->mesh_row_count, h->mesh_row_offset, h->mesh_group_base,
                                    h->mesh_desc + base, h->mesh_verts + 9 * base,
                                    with_color ? h->mesh_colors + 3 * base : nullptr, n_piece, (unsigned*)(d_total + 1)));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->mesh_total[1])
            return fail(h, TSDF_E_HALO, "tsdf_mesh_extract: a vertex colour reads outside the stored layers (halo >= 1 needed)");
        base += n_piece;
    }
    h->mesh_ntri = (int64_t)n;
    h->mesh_has_color = with_color != 0;
    if (n_triangles) *n_triangles = (int64_t)n;
    return TSDF_OK;
}

int tsdf_mesh_read(tsdf_handle* h, float* vertices, float* colors, int64_t capacity_triangles) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (h->mesh_ntri < 0) return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: call tsdf_mesh_extract first");
    if (!vertices && h->mesh_ntri > 0) return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: vertices is NULL");
    if (capacity_triangles < h->mesh_ntri)
        return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: room for %lld triangles, the mesh has %lld",
                    (long long)capacity_triangles, (long long)h->mesh_ntri);
    if (colors && !h->mesh_has_color) return fail(h, TSDF_E_BADARG, "tsdf_mesh_read: the last extraction had no colours");
    const size_t n = (size_t)h->mesh_ntri;
    if (n == 0) return TSDF_OK;
    HIP_TRY(h, hipMemcpyAsync(vertices, h->mesh_verts, n * 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (colors) HIP_TRY(h, hipMemcpyAsync(colors, h->mesh_colors, n * 12 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return TSDF_OK;
}

int tsdf_mesh_device(tsdf_handle* h, const float** vertices, const float** colors, int64_t* n_triangles) {
    if (!h) return TSDF_E_BADARG;
    if (h->mesh_ntri < 0) return fail(h, TSDF_E_BADARG, "tsdf_mesh_device: call tsdf_mesh_extract first");
    if (vertices) *vertices = h->mesh_verts;
    if (colors) *colors = h->mesh_has_color ? reinterpret_cast<const float*>(h->mesh_colors) : nullptr;
    if (n_triangles) *n_triangles = h->mesh_ntri;
    return TSDF_OK;
}

// ---- volume I/O --------------------------------------------------------------------------------------

namespace {

// Copy `planes` float arrays of `n` voxels between host and the interleaved device layout, through a
// bounded device scratch (chunked so a 2048^3 slab does not need a second copy of itself).
int volume_io(tsdf_handle* h, bool download, bool color, int64_t first, int64_t n, float* const* host) {
    const int planes = color ? 4 : 2;
    const int64_t chunk = (int64_t)1 << 24;     // 16 Mi voxels per pass
    float* scratch = nullptr;
    const int64_t cap = n < chunk ? n : chunk;
    HIP_TRY(h, hipMalloc((void**)&scratch, (size_t)cap * planes * sizeof(float)));
    hipError_t e = hipSuccess;
    for (int64_t off = 0; off < n && e == hipSuccess; off += chunk) {
        const int64_t c = (n - off) < chunk ? (n - off) : chunk;
        float* pl[4] = {scratch, scratch + cap, scratch + 2 * cap, scratch + 3 * cap};
        if (download) {
            if (color) e = launch_split4(h->stream, h->crgb + first + off, pl[0], pl[1], pl[2], pl[3], c);
            else e = launch_split(h->stream, h->dw + first + off, pl[0], pl[1], c);
            for (int q = 0; q < planes && e == hipSuccess; ++q)
                e = hipMemcpyAsync(host[q] + off, pl[q], (size_t)c * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        } else {
            for (int q = 0; q < planes && e == hipSuccess; ++q)
                e = hipMemcpyAsync(pl[q], host[q] + off, (size_t)c * sizeof(float), hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) {
                if (color) e = launch_merge4(h->stream, h->crgb + first + off, pl[0], pl[1], pl[2], pl[3], c);
                else e = launch_merge(h->stream, h->dw + first + off, pl[0], pl[1], c);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(scratch);
    if (e != hipSuccess) return fail(h, TSDF_E_HIP, "volume I/O: %s", hipGetErrorString(e));
    return TSDF_OK;
}

// the handle's OWN layers, in increasing x: one range of a plain slab, one per block of a block-cyclic handle
int owned_io(tsdf_handle* h, bool download, bool color, float* const* host) {
    const Grid& g = h->grid;
    const int64_t mm = (int64_t)g.m * g.m;
    if (g.blk_own <= 0) return volume_io(h, download, color, (g.own_x0 - g.xs) * mm, (g.own_x1 - g.own_x0) * mm, host);
    const int planes = color ? 4 : 2;
    const int64_t n = (int64_t)g.blk_own * mm;
    for (int b = 0; b < g.n_blocks; ++b) {
        float* at[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int q = 0; q < planes; ++q) at[q] = host[q] + (int64_t)b * n;
        const int rc = volume_io(h, download, color, ((int64_t)b * g.blk_layers + (g.own_x0 - g.blk_first)) * mm, n, at);
        if (rc) return rc;
    }
    return TSDF_OK;
}

}  // namespace

int tsdf_download(tsdf_handle* h, float* D, float* W) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!D || !W) return fail(h, TSDF_E_BADARG, "tsdf_download: null output");
    float* host[4] = {D, W, nullptr, nullptr};
    return owned_io(h, true, false, host);
}

int tsdf_upload(tsdf_handle* h, const float* D, const float* W) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!D || !W) return fail(h, TSDF_E_BADARG, "tsdf_upload: null input");
    float* host[4] = {const_cast<float*>(D), const_cast<float*>(W), nullptr, nullptr};
    return owned_io(h, false, false, host);
}

int tsdf_upload_with_halo(tsdf_handle* h, const float* D, const float* W) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!D || !W) return fail(h, TSDF_E_BADARG, "tsdf_upload_with_halo: null input");
    if ((rc = refuse_cyclic(h, "tsdf_upload_with_halo")) != TSDF_OK) return rc;
    float* host[4] = {const_cast<float*>(D), const_cast<float*>(W), nullptr, nullptr};
    return volume_io(h, false, false, 0, h->n_stored, host);
}

int tsdf_download_color(tsdf_handle* h, float* Color_W, float* R, float* G, float* B) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!h->crgb) return fail(h, TSDF_E_BADARG, "volume was created with with_color=0");
    if (!Color_W || !R || !G || !B) return fail(h, TSDF_E_BADARG, "tsdf_download_color: null output");
    float* host[4] = {Color_W, R, G, B};
    return owned_io(h, true, true, host);
}

int tsdf_upload_color(tsdf_handle* h, const float* Color_W, const float* R, const float* G, const float* B) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!h->crgb) return fail(h, TSDF_E_BADARG, "volume was created with with_color=0");
    if (!Color_W || !R || !G || !B) return fail(h, TSDF_E_BADARG, "tsdf_upload_color: null input");
    float* host[4] = {const_cast<float*>(Color_W), const_cast<float*>(R), const_cast<float*>(G), const_cast<float*>(B)};
    return owned_io(h, false, true, host);
}

int tsdf_upload_color_with_halo(tsdf_handle* h, const float* Color_W, const float* R, const float* G, const float* B) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if ((rc = refuse_cyclic(h, "tsdf_upload_color_with_halo")) != TSDF_OK) return rc;
    if (!h->crgb) return fail(h, TSDF_E_BADARG, "volume was created with with_color=0");
    if (!Color_W || !R || !G || !B) return fail(h, TSDF_E_BADARG, "tsdf_upload_color_with_halo: null input");
    float* host[4] = {const_cast<float*>(Color_W), const_cast<float*>(R), const_cast<float*>(G), const_cast<float*>(B)};
    return volume_io(h, false, true, 0, h->n_stored, host);
}

// ---- checkpoint ----------------------------------------------------------------------------------------

namespace {
struct VolHeader {                 // TSDFVOL2, little-endian, 80 bytes
    char magic[8];
    int32_t m, x0, x1, has_color;  // x0, x1: the slab the writer owned (informative)
    float width, height, depth, delta, epsilon;
    int32_t xs;                    // the file holds the x layers [xs, xe): the writer's slab AND its halo
    double origin[3];
    int32_t xe, stride;            // stride > 0: a block-cyclic handle's file -- EVERY stored layer of every block, as they lie in
                                   // memory ([xs, xe) is block 0's stored range, unclipped; blocks at x0 + b * stride); 0: one slab
};
static_assert(sizeof(VolHeader) == 80, "checkpoint header layout");

bool read_plane(FILE* f, long long plane_floats, int plane, long long first, float* dst, size_t n) {
    const long long off = (long long)sizeof(VolHeader) + ((long long)plane * plane_floats + first) * (long long)sizeof(float);
    if (fseeko(f, (off_t)off, SEEK_SET) != 0) return false;
    return std::fread(dst, sizeof(float), n, f) == n;
}
}  // namespace

int tsdf_save(tsdf_handle* h, const char* path) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!path) return fail(h, TSDF_E_BADARG, "tsdf_save: null path");
    const size_t n = (size_t)h->n_stored;              // slab + halo: a restored shard needs its halo layers too
    std::vector<float> buf;
    try { buf.resize(n * 4); } catch (...) { return fail(h, TSDF_E_NOMEM, "tsdf_save: out of host memory"); }
    VolHeader hd;
    std::memset(&hd, 0, sizeof hd);
    std::memcpy(hd.magic, "TSDFVOL2", 8);
    hd.m = h->grid.m; hd.x0 = h->grid.own_x0; hd.x1 = h->grid.own_x1; hd.has_color = h->crgb ? 1 : 0;
    hd.xs = h->grid.xs; hd.xe = h->grid.xe;
    if (h->grid.blk_own > 0) { hd.xs = h->grid.blk_first; hd.xe = h->grid.blk_first + h->grid.blk_layers; hd.stride = h->grid.blk_stride; }
    hd.width = h->cfg.width; hd.height = h->cfg.height; hd.depth = h->cfg.depth;
    hd.delta = h->cfg.delta; hd.epsilon = h->cfg.epsilon;
    std::memcpy(hd.origin, h->cfg.origin, sizeof hd.origin);
    // written next to the destination and renamed over it once complete: a failed save leaves neither a truncated file
    // with a valid header nor a destroyed earlier checkpoint
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(h, TSDF_E_BADARG, "tsdf_save: cannot open %s", tmp.c_str());
    bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1;
    float* host[4] = {buf.data(), buf.data() + n, buf.data() + 2 * n, buf.data() + 3 * n};
    if (ok) {
        rc = volume_io(h, true, false, 0, (int64_t)n, host);
        ok = rc == TSDF_OK && std::fwrite(buf.data(), sizeof(float), 2 * n, f) == 2 * n;
    }
    if (ok && h->crgb) {
        rc = volume_io(h, true, true, 0, (int64_t)n, host);
        ok = rc == TSDF_OK && std::fwrite(buf.data(), sizeof(float), 4 * n, f) == 4 * n;
    }
    ok = (std::fflush(f) == 0) && ok;
    ok = (std::fclose(f) == 0) && ok;
    if (rc || !ok) {
        std::remove(tmp.c_str());
        return rc ? rc : fail(h, TSDF_E_BADARG, "tsdf_save: short write to %s", tmp.c_str());
    }
    if (std::rename(tmp.c_str(), path) != 0) {
        std::remove(tmp.c_str());
        return fail(h, TSDF_E_BADARG, "tsdf_save: cannot rename %s to %s", tmp.c_str(), path);
    }
    return TSDF_OK;
}

int tsdf_load(tsdf_handle* h, const char* path) {
    int rc = check_ready(h, false);
    if (rc) return rc;
    if (!path) return fail(h, TSDF_E_BADARG, "tsdf_load: null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(h, TSDF_E_BADARG, "tsdf_load: cannot open %s", path);
    VolHeader hd;
    if (std::fread(&hd, sizeof hd, 1, f) != 1 || std::memcmp(hd.magic, "TSDFVOL2", 8) != 0) {
        std::fclose(f);
        return fail(h, TSDF_E_BADARG, "tsdf_load: %s is not a TSDFVOL2 file", path);
    }
    const Grid& g = h->grid;
    if (hd.m != g.m || (hd.has_color != 0) != (h->crgb != nullptr)) {
        std::fclose(f);
        return fail(h, TSDF_E_BADARG, "tsdf_load: file holds m=%d colour=%d, handle has m=%d colour=%d", hd.m, hd.has_color,
                    g.m, h->crgb ? 1 : 0);
    }
    // the geometry the voxel values were fused under must be the handle's (a D value means nothing under another delta)
    if (hd.width != h->cfg.width || hd.height != h->cfg.height || hd.depth != h->cfg.depth || hd.delta != h->cfg.delta ||
        hd.epsilon != h->cfg.epsilon || std::memcmp(hd.origin, h->cfg.origin, sizeof hd.origin) != 0) {
        std::fclose(f);
        return fail(h, TSDF_E_BADARG, "tsdf_load: %s was written for another volume (extent %gx%gx%g, origin %g %g %g, delta %g, epsilon %g)",
                    path, (double)hd.width, (double)hd.height, (double)hd.depth, hd.origin[0], hd.origin[1], hd.origin[2],
                    (double)hd.delta, (double)hd.epsilon);
    }
    // every STORED layer of this handle (slab and halo) must come from the file: a halo left at its old contents
    // would silently break the 'halo == neighbour's interior' invariant the sharded tracker relies on
    const long long mm = (long long)g.m * g.m;
    // What goes where: pieces {first float of the piece in a plane of the file, first voxel in the handle, voxels}.
    struct Piece { long long file_first, local_first, count; };
    std::vector<Piece> pieces;
    long long plane_floats = 0;
    try {
        if (hd.stride > 0) {
            // a block-cyclic handle's own file: the same placement, read back as it was written
            if (g.blk_own <= 0 || hd.stride != g.blk_stride || hd.x0 != g.own_x0 || hd.x1 != g.own_x1 || hd.xs != g.blk_first ||
                hd.xe != g.blk_first + g.blk_layers) {
                std::fclose(f);
                return fail(h, TSDF_E_HALO, "tsdf_load: %s holds the blocks [%d,%d) + %d j with the stored range [%d,%d) of a block-cyclic handle; this handle is placed otherwise",
                            path, hd.x0, hd.x1, hd.stride, hd.xs, hd.xe);
            }
            plane_floats = (long long)h->n_stored;
            pieces.push_back({0, 0, (long long)h->n_stored});
        } else {
            if (hd.xs < 0 || hd.xe > hd.m || hd.xs >= hd.xe) { std::fclose(f); return fail(h, TSDF_E_BADARG, "tsdf_load: %s: bad layer range [%d,%d)", path, hd.xs, hd.xe); }
            plane_floats = (long long)(hd.xe - hd.xs) * mm;
            const int nb = g.blk_own > 0 ? g.n_blocks : 1;
            for (int b = 0; b < nb; ++b) {
                // the stored layers of the slab / of block b that lie in the grid
                const int first = g.blk_own > 0 ? g.blk_first + b * g.blk_stride : g.xs;
                const int lo = first < 0 ? 0 : first;
                const int hi = g.blk_own > 0 ? (first + g.blk_layers > g.m ? g.m : first + g.blk_layers) : g.xe;
                if (hd.xs > lo || hd.xe < hi) {
                    std::fclose(f);
                    return fail(h, TSDF_E_HALO, "tsdf_load: file holds x layers [%d,%d), this handle stores [%d,%d) (slab [%d,%d) + halo %d)",
                                hd.xs, hd.xe, lo, hi, g.own_x0 + (g.blk_own > 0 ? b * g.blk_stride : 0), g.own_x1 + (g.blk_own > 0 ? b * g.blk_stride : 0), h->cfg.halo);
                }
                pieces.push_back({(long long)(lo - hd.xs) * mm, ((long long)(g.blk_own > 0 ? b * g.blk_layers : 0) + (lo - first)) * mm, (long long)(hi - lo) * mm});
            }
        }
    } catch (...) { std::fclose(f); return fail(h, TSDF_E_NOMEM, "tsdf_load: out of host memory"); }
    size_t n = 0;
    for (const Piece& pc : pieces) if ((size_t)pc.count > n) n = (size_t)pc.count;
    {
        // the whole file must be there BEFORE anything is uploaded: a file cut inside its colour part must not leave the
        // handle with new D / W and old colour
        const long long want = (long long)sizeof(VolHeader) + plane_floats * (long long)sizeof(float) * (hd.has_color ? 6 : 2);
        long long have = -1;
        if (fseeko(f, 0, SEEK_END) == 0) have = (long long)ftello(f);
        if (have < want || fseeko(f, (off_t)sizeof(VolHeader), SEEK_SET) != 0) {
            std::fclose(f);
            return fail(h, TSDF_E_BADARG, "tsdf_load: %s is truncated (%lld of %lld bytes); nothing was loaded", path, have, want);
        }
    }
    std::vector<float> buf;
    try { buf.resize(n * 4); } catch (...) { std::fclose(f); return fail(h, TSDF_E_NOMEM, "tsdf_load: out of host memory"); }
    float* host[4] = {buf.data(), buf.data() + n, buf.data() + 2 * n, buf.data() + 3 * n};
    bool ok = true;
    for (size_t i = 0; i < pieces.size() && ok && rc == TSDF_OK; ++i) {
        const Piece& pc = pieces[i];
        ok = read_plane(f, plane_floats, 0, pc.file_first, host[0], (size_t)pc.count) && read_plane(f, plane_floats, 1, pc.file_first, host[1], (size_t)pc.count);
        if (ok) rc = volume_io(h, false, false, pc.local_first, pc.count, host);
    }
    for (size_t i = 0; i < pieces.size() && ok && rc == TSDF_OK && h->crgb; ++i) {
        const Piece& pc = pieces[i];
        for (int q = 0; q < 4 && ok; ++q) ok = read_plane(f, plane_floats, 2 + q, pc.file_first, host[q], (size_t)pc.count);
        if (ok) rc = volume_io(h, false, true, pc.local_first, pc.count, host);
    }
    std::fclose(f);
    if (!ok) return fail(h, TSDF_E_BADARG, "tsdf_load: %s is truncated", path);
    return rc;
}
