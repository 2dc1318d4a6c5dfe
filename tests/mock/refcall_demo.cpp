// refcall_demo.cpp -- the frame loop of tools/shim_demo.cpp once more, but through the reference's EXACT signatures
// (hotpath.hpp with -DTSDF_WITH_EIGEN_PCL -DTSDF_WITH_ROS, against the mock Eigen / PCL / ROS headers of this
// directory): Eigen::Vector3d& origin, pcl::PointCloud<...>::Ptr clouds, camera_info_cb, Eigen-typed trans / rot.
// The GPU test runs both demos on the same frame dump and requires identical trajectory files.
// Usage: refcall_demo frames.bin m trajectory.txt      (frame dump: tools/dump_frames.py)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "sdf_3d_reconstruction/hotpath.hpp"

using namespace Eigen;

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s frames.bin m trajectory.txt\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    int32_t hdr[3];
    sensor_msgs::CameraInfo info;
    if (std::fread(hdr, sizeof hdr, 1, f) != 1 || std::fread(info.K, sizeof(double), 9, f) != 9) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2];
    const bool normals_early = argc > 4 && std::string(argv[4]) == "normals-at-track";
    try {
        Vector3d sdf_origin(-3.0, -3.0, -0.5);
        SDF* sdf = new SDF(std::atoi(argv[2]), 6.0, 6.0, 3.5, sdf_origin, 0.3, 0.025);          // sdf_reconstruction.cpp:85
        CameraTracking* camera_tracking = new CameraTracking(20, 0.001, 1.0, 0.01, sdf);         // :88
        camera_tracking->camera_info_cb(sensor_msgs::CameraInfoConstPtr(new sensor_msgs::CameraInfo(info)));
        FILE* out = std::fopen(argv[3], "w");
        // all frames first, as the clouds kinect_callback holds (pcl::PointCloud<...>::Ptr in pageable memory): the loop
        // below then contains nothing but the reference's two hot calls and is timed as a whole
        std::vector<double> stamps((size_t)n);
        std::vector<pcl::PointCloud<pcl::PointXYZRGB>::Ptr> clouds;
        std::vector<pcl::PointCloud<pcl::Normal>::Ptr> normal_clouds;
        {
            std::vector<float> xyz((size_t)w * h * 3), nrm((size_t)w * h * 3);
            std::vector<uint8_t> rgb((size_t)w * h * 3);
            for (int k = 0; k < n; ++k) {
                if (std::fread(&stamps[k], sizeof(double), 1, f) != 1 || std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() ||
                    std::fread(nrm.data(), 4, nrm.size(), f) != nrm.size() || std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size())
                    return 2;
                pcl::PointCloud<pcl::PointXYZRGB>::Ptr cloud_filtered(new pcl::PointCloud<pcl::PointXYZRGB>);
                pcl::PointCloud<pcl::Normal>::Ptr normals(new pcl::PointCloud<pcl::Normal>);
                cloud_filtered->width = normals->width = (uint32_t)w;
                cloud_filtered->height = normals->height = (uint32_t)h;
                cloud_filtered->points.resize((size_t)w * h);
                normals->points.resize((size_t)w * h);
                for (size_t i = 0; i < (size_t)w * h; ++i) {
                    pcl::PointXYZRGB& p = cloud_filtered->points[i];
                    p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2];
                    p.r = rgb[3 * i]; p.g = rgb[3 * i + 1]; p.b = rgb[3 * i + 2]; p.a = 255;
                    pcl::Normal& q = normals->points[i];
                    q.normal_x = nrm[3 * i]; q.normal_y = nrm[3 * i + 1]; q.normal_z = nrm[3 * i + 2]; q.curvature = 0.f;
                }
                clouds.push_back(cloud_filtered); normal_clouds.push_back(normals);
            }
        }
        const int warm = n > 8 ? 5 : 0;                       // frames in front of the timed ones (library threads, pinned buffers)
        std::chrono::steady_clock::time_point t0;
        for (int frame_num = 1; frame_num <= n; ++frame_num) {
            if (frame_num == 2 + warm) { tsdf_synchronize(sdf->handle()); t0 = std::chrono::steady_clock::now(); }
            const pcl::PointCloud<pcl::PointXYZRGB>::Ptr& cloud_filtered = clouds[frame_num - 1];
            const pcl::PointCloud<pcl::Normal>::Ptr& normals = normal_clouds[frame_num - 1];
            if (frame_num > 1) {                                                                  // :69-72
                if (normals_early) camera_tracking->estimate_new_position(sdf, cloud_filtered, normals);   // one more argument than :70
                else camera_tracking->estimate_new_position(sdf, cloud_filtered);
                const Eigen::Vector3d& trans = camera_tracking->trans;
                std::fprintf(out, "%.4f %.4f %.4f %.4f\n", stamps[frame_num - 1], trans.x(), trans.y(), trans.z());
            }
            sdf->update(camera_tracking, cloud_filtered, normals);                                // :74
        }
        tsdf_synchronize(sdf->handle());
        if (n - 1 - warm > 0) {
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "RATE %.1f frames/s over %d frames of %dx%d through estimate_new_position + update (exact reference signatures%s)\n",
                         (double)(n - 1 - warm) / sec, n - 1 - warm, w, h, normals_early ? "; the normals handed to estimate_new_position as well" : "");
        }
        std::fclose(out);
        std::printf("final pose t = %.9f %.9f %.9f  rot00 = %.9f\n", camera_tracking->trans(0), camera_tracking->trans(1),
                    camera_tracking->trans(2), camera_tracking->rot(0, 0));
        delete camera_tracking;
        delete sdf;
    } catch (const tsdf_shim::Error& e) {
        std::fprintf(stderr, "tsdf error %d: %s\n", e.code, e.what());
        return 1;
    }
    std::fclose(f);
    return 0;
}
