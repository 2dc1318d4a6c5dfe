// sdf_offline.cpp -- offline C++ host of the hot path for TUM RGB-D directories: the reference's frame loop
// (sdf_reconstruction.cpp:21-110) without ROS.  Reads <dir>/depth.txt and the 16-bit depth PNGs it lists
// (metres = value / 5000), pre-processes each image on the GPU (tsdf_set_depth_frame), tracks, appends the pose
// to a TUM-format trajectory file (sdf_reconstruction.cpp:4-17) and integrates.
//
//   sdf_offline <tum_dir> <voxels> <trajectory.txt> [max_frames] [fx fy cx cy] [bilateral_radius] [mesh.ply] [groundtruth.txt]
//
// bilateral_radius: 0 = no depth filter, > 0 = bilateral grid (the default), < 0 = exact windowed filter of radius |r|.
// mesh.ply ("-" = none): the visualiser's mesh of the final volume.  groundtruth.txt (TUM format: stamp tx ty tz qx qy qz
// qw, camera -> world): the reference's _useGroundTruth mode (sdf_reconstruction.cpp:51-66) -- no tracking, every depth
// frame is fused at the ground-truth pose nearest in time (frames without a pose within 20 ms are skipped).
//
// Only zlib is needed (minimal PNG reader below: 8/16-bit greyscale, non-interlaced, all five filters).
#include <zlib.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../include/sdf_3d_reconstruction/hotpath.hpp"

using namespace tsdf_shim;

static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// 16-bit (or 8-bit) greyscale PNG -> uint16 image.  Returns false on anything unexpected.
static bool read_png_gray16(const std::string& path, std::vector<uint16_t>& img, int& w, int& h) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat;
    w = h = 0;
    while (pos + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[pos]);
        const char* type = (const char*)&buf[pos + 4];
        if (pos + 12 + len > buf.size()) return false;
        const unsigned char* data = &buf[pos + 8];
        if (!std::strncmp(type, "IHDR", 4)) {
            w = (int)be32(data); h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!std::strncmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::strncmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || ctype != 0 || interlace != 0 || (depth != 16 && depth != 8)) return false;
    const int bpp = depth / 8;
    const size_t stride = (size_t)w * bpp;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return false;
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    img.resize((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const unsigned char* line = &raw[(stride + 1) * y];
        const int ft = line[0];
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= (size_t)bpp ? prev[x - bpp] : 0;
            int v = line[1 + x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: return false;
            }
            cur[x] = (unsigned char)v;
        }
        for (int x = 0; x < w; ++x)
            img[(size_t)y * w + x] = bpp == 2 ? (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]) : cur[x];
        prev.swap(cur);
    }
    return true;
}

// Eigen::Quaterniond(Matrix3d) for the pose file
static void quat_from_rot(const Mat3& R, double q[4] /*x y z w*/) {
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q[3] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(std::fmax(R[4 * i] - R[4 * j] - R[4 * k] + 1.0, 0.0));
        q[i] = 0.5 * s; s = s > 0 ? 0.5 / s : 0.0;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
}

struct GtPose { double stamp, t[3], q[4]; };

static std::vector<GtPose> read_groundtruth(const std::string& path) {
    std::vector<GtPose> out;
    std::ifstream f(path);
    for (std::string line; std::getline(f, line);) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        GtPose g;
        if (ss >> g.stamp >> g.t[0] >> g.t[1] >> g.t[2] >> g.q[0] >> g.q[1] >> g.q[2] >> g.q[3]) out.push_back(g);
    }
    return out;
}

// pose nearest in time (poses sorted by stamp, as TUM files are); -1 if none within max_dt
static int nearest_pose(const std::vector<GtPose>& gt, double stamp, double max_dt) {
    size_t lo = 0, hi = gt.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (gt[mid].stamp < stamp) lo = mid + 1; else hi = mid; }
    int best = -1;
    double bd = max_dt;
    for (size_t c = lo > 0 ? lo - 1 : 0; c < gt.size() && c <= lo; ++c) {
        const double d = std::fabs(gt[c].stamp - stamp);
        if (d <= bd) { bd = d; best = (int)c; }
    }
    return best;
}

static Mat3 rot_from_quat(const double q[4] /*x y z w*/) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    return Mat3{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
}

// Binary little-endian PLY triangle soup; vertices moved to the world frame as SDF::visualize does (sdf.cpp:355-369).
static bool write_ply(const char* path, const std::vector<float>& verts, const Vec3& origin) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const size_t nv = verts.size() / 3, nt = nv / 3;
    std::fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %zu\nproperty float x\nproperty float y\n"
                    "property float z\nelement face %zu\nproperty list uchar int vertex_indices\nend_header\n", nv, nt);
    for (size_t v = 0; v < nv; ++v) {
        const float p[3] = {(float)(verts[3 * v] + origin[0]), (float)(verts[3 * v + 1] + origin[1]),
                            (float)(verts[3 * v + 2] + origin[2])};
        std::fwrite(p, sizeof(float), 3, f);
    }
    for (size_t t = 0; t < nt; ++t) {
        const unsigned char three = 3;
        const int32_t idx[3] = {(int32_t)(3 * t), (int32_t)(3 * t + 1), (int32_t)(3 * t + 2)};
        std::fwrite(&three, 1, 1, f);
        std::fwrite(idx, sizeof(int32_t), 3, f);
    }
    return std::fclose(f) == 0;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <tum_dir> <voxels> <trajectory.txt> [max_frames] [fx fy cx cy] [bilateral_radius] [mesh.ply|-] [groundtruth.txt]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const int m = std::atoi(argv[2]);
    const int max_frames = argc > 4 ? std::atoi(argv[4]) : 0;
    Mat3 K{525.0, 0, 319.5, 0, 525.0, 239.5, 0, 0, 1.0};     // ROS default of the TUM bags
    if (argc > 8) { K[0] = std::atof(argv[5]); K[4] = std::atof(argv[6]); K[2] = std::atof(argv[7]); K[5] = std::atof(argv[8]); }
    tsdf_preproc_params pp;
    tsdf_default_preproc(&pp);
    if (argc > 9) {                                            // 0: no filter; > 0: bilateral grid; < 0: windowed filter of that radius
        const int r = std::atoi(argv[9]);
        pp.radius = r < 0 ? -r : r;
        pp.grid_filter = r < 0 ? 0 : 1;
    }

    std::ifstream list(dir + "/depth.txt");
    if (!list) { std::fprintf(stderr, "cannot open %s/depth.txt\n", dir.c_str()); return 2; }
    std::vector<std::pair<double, std::string>> items;
    for (std::string line; std::getline(list, line);) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        double stamp; std::string name;
        if (ss >> stamp >> name) items.emplace_back(stamp, name);
    }
    if (max_frames > 0 && (int)items.size() > max_frames) items.resize(max_frames);
    std::vector<GtPose> gt;
    if (argc > 11) {
        gt = read_groundtruth(argv[11]);
        if (gt.empty()) { std::fprintf(stderr, "no poses in %s\n", argv[11]); return 2; }
    }
    const bool fusion_only = !gt.empty();

    try {
        const Vec3 origin{-3.0, -3.0, -0.5};
        tsdf_config cfg;
        tsdf_default_config(&cfg);
        cfg.with_color = 0;                                   // depth-only input
        SDF sdf(m, 6.0f, 6.0f, 3.5f, origin, 0.3f, 0.025f, &cfg);
        CameraTracking tracker(20, 0.001f, 1.0f, 0.01f, &sdf);
        tracker.set_K(K);
        FILE* out = std::fopen(argv[3], "w");
        if (!out) { std::perror(argv[3]); return 2; }
        std::vector<uint16_t> depth;
        int frame_num = 0, lost = 0, skipped = 0;
        double hot = 0.0;
        for (const auto& it : items) {
            int w = 0, h = 0;
            if (!read_png_gray16(dir + "/" + it.second, depth, w, h)) {
                std::fprintf(stderr, "skipping unreadable %s\n", it.second.c_str());
                continue;
            }
            int gi = -1;
            if (fusion_only && (gi = nearest_pose(gt, it.first, 0.02)) < 0) { ++skipped; continue; }
            ++frame_num;
            const auto t0 = std::chrono::steady_clock::now();
            sdf.set_depth_frame(depth.data(), nullptr, w, h, &pp);
            if (fusion_only) {                                                    // sdf_reconstruction.cpp:51-66
                const Vec3 tr{gt[gi].t[0], gt[gi].t[1], gt[gi].t[2]};
                tracker.set_camera_transformation(rot_from_quat(gt[gi].q), tr);
            } else if (frame_num > 1) {                                                  // sdf_reconstruction.cpp:69-72
                try { tracker.estimate_new_position(&sdf); }
                catch (const Error& e) { ++lost; std::fprintf(stderr, "frame %d: %s\n", frame_num, e.what()); }
                double q[4];
                quat_from_rot(tracker.rot, q);
                std::fprintf(out, "%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n", it.first, tracker.trans[0], tracker.trans[1],
                             tracker.trans[2], q[0], q[1], q[2], q[3]);
            }
            sdf.update(&tracker);                                                 // :74
            tsdf_synchronize(sdf.handle());
            hot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        std::fclose(out);
        long long n_tri = -1;
        if (argc > 10 && std::strcmp(argv[10], "-") != 0) {                       // the visualiser's mesh, once, at the end
            std::vector<float> verts;
            n_tri = sdf.mesh(verts);
            if (!write_ply(argv[10], verts, origin)) { std::perror(argv[10]); return 2; }
        }
        std::printf("{\"mesh_triangles\": %lld, \"fusion_only\": %s, \"frames_without_pose\": %d}\n", n_tri,
                    fusion_only ? "true" : "false", skipped);
        std::printf("{\"frames\": %d, \"track_errors\": %d, \"fps_incl_upload_and_preprocessing\": %.1f, "
                    "\"final_t\": [%.9f, %.9f, %.9f]}\n", frame_num, lost, frame_num / hot,
                    tracker.trans[0], tracker.trans[1], tracker.trans[2]);
    } catch (const Error& e) {
        std::fprintf(stderr, "tsdf error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
