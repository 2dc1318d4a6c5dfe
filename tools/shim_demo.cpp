// shim_demo.cpp -- the reference's frame loop (sdf_reconstruction.cpp:21-110) written against the C++
// shim: frame 1 is integrated at the hard-coded initial pose, every later frame is tracked, its pose
// appended to a TUM-format trajectory file (timestamp tx ty tz qx qy qz qw, 4 decimals,
// sdf_reconstruction.cpp:4-17), then integrated.
//
// Input: a raw frame dump written by tools/dump_frames.py:
//   header  int32 n_frames, width, height; double K[9]
//   per frame: double stamp; float xyz[h*w*3]; float nrm[h*w*3]; uint8 rgb[h*w*3]
// Usage: shim_demo frames.bin m trajectory.txt
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/sdf_3d_reconstruction/hotpath.hpp"

using namespace tsdf_shim;

// Eigen::Quaterniond(Matrix3d) (Shepperd's branches) for the pose file.
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
        double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * s; s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s frames.bin m trajectory.txt\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    int32_t hdr[3];
    Mat3 K;
    if (std::fread(hdr, sizeof hdr, 1, f) != 1 || std::fread(K.data(), sizeof(double), 9, f) != 9) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2];
    const int m = std::atoi(argv[2]);
    try {
        const Vec3 origin{-3.0, -3.0, -0.5};
        SDF sdf(m, 6.0f, 6.0f, 3.5f, origin, 0.3f, 0.025f);                  // sdf_reconstruction.cpp:83-85
        CameraTracking tracker(20, 0.001f, 1.0f, 0.01f, &sdf);                // :88
        tracker.set_K(K);
        FILE* out = std::fopen(argv[3], "w");
        std::vector<float> xyz((size_t)w * h * 3), nrm((size_t)w * h * 3);
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        for (int frame_num = 1; frame_num <= n; ++frame_num) {
            double stamp;
            if (std::fread(&stamp, sizeof stamp, 1, f) != 1 || std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() ||
                std::fread(nrm.data(), 4, nrm.size(), f) != nrm.size() || std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size())
                return 2;
            const OrganizedCloud cloud{xyz.data(), rgb.data(), w, h};
            const NormalCloud normals{nrm.data(), w, h};
            if (frame_num > 1) {                                              // :69-72
                tracker.estimate_new_position(&sdf, cloud);
                double q[4];
                quat_from_rot(tracker.rot, q);
                std::fprintf(out, "%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n", stamp, tracker.trans[0], tracker.trans[1],
                             tracker.trans[2], q[0], q[1], q[2], q[3]);
            }
            sdf.update(&tracker, cloud, normals);                             // :74
        }
        std::fclose(out);
        std::printf("final pose t = %.9f %.9f %.9f\n", tracker.trans[0], tracker.trans[1], tracker.trans[2]);
    } catch (const Error& e) {
        std::fprintf(stderr, "tsdf error %d: %s\n", e.code, e.what());
        return 1;
    }
    std::fclose(f);
    return 0;
}
