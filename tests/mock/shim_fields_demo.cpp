// shim_fields_demo.cpp -- what the exact-type shim must get right beyond the statements of sdf_reconstruction.cpp
// (hotpath.hpp with -DTSDF_WITH_EIGEN_PCL, mock Eigen / PCL headers of this directory; run by tests/test_cpp_shim.py on the GPU):
//   1. a pose ASSIGNED to the public rot / trans fields (plain members in the reference, camera_tracking.h:43-59) reaches
//      the integration exactly like set_camera_transformation does;
//   2. CameraTracking(max_iter, max_twist, v_h, w_h, sdf) with other constants than the SDF's reconfigures the handle;
//   3. a cloud changed IN PLACE between estimate_new_position and update is uploaded again (no stale points);
//   4. the rest of the two classes' public surface (sdf.h:113-181, camera_tracking.h:69-101): index / coordinate maps,
//      projections and get_partial_derivative -- printed as "H ..." lines that the test compares with the oracle's.
// Usage: shim_fields_demo frames.bin m        prints "ok" lines; exit code 0 = all three hold
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sdf_3d_reconstruction/hotpath.hpp"

using namespace Eigen;
typedef pcl::PointCloud<pcl::PointXYZRGB>::Ptr CloudPtr;
typedef pcl::PointCloud<pcl::Normal>::Ptr NormalsPtr;

static bool same_volume(SDF* a, SDF* b) {
    std::vector<float> Da, Wa, Db, Wb;
    a->download(Da, Wa); b->download(Db, Wb);
    return Da.size() == Db.size() && !std::memcmp(Da.data(), Db.data(), Da.size() * 4) && !std::memcmp(Wa.data(), Wb.data(), Wa.size() * 4);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[3];
    sensor_msgs::CameraInfo info;
    if (std::fread(hdr, sizeof hdr, 1, f) != 1 || std::fread(info.K, sizeof(double), 9, f) != 9) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2], m = std::atoi(argv[2]);
    if (n < 2) return 2;
    std::vector<CloudPtr> clouds; std::vector<NormalsPtr> nrms;
    for (int k = 0; k < 2; ++k) {
        double stamp;
        std::vector<float> xyz((size_t)w * h * 3), nrm((size_t)w * h * 3);
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        if (std::fread(&stamp, 8, 1, f) != 1 || std::fread(xyz.data(), 4, xyz.size(), f) != xyz.size() ||
            std::fread(nrm.data(), 4, nrm.size(), f) != nrm.size() || std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) return 2;
        CloudPtr c(new pcl::PointCloud<pcl::PointXYZRGB>); NormalsPtr q(new pcl::PointCloud<pcl::Normal>);
        c->width = q->width = (uint32_t)w; c->height = q->height = (uint32_t)h;
        c->points.resize((size_t)w * h); q->points.resize((size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            pcl::PointXYZRGB& p = c->points[i];
            p.x = xyz[3 * i]; p.y = xyz[3 * i + 1]; p.z = xyz[3 * i + 2]; p.data3 = 1.f;
            p.r = rgb[3 * i]; p.g = rgb[3 * i + 1]; p.b = rgb[3 * i + 2]; p.a = 255;
            pcl::Normal& nn = q->points[i];
            nn.normal_x = nrm[3 * i]; nn.normal_y = nrm[3 * i + 1]; nn.normal_z = nrm[3 * i + 2]; nn.data_n3 = 0.f; nn.curvature = 0.f;
        }
        clouds.push_back(c); nrms.push_back(q);
    }
    std::fclose(f);
    int bad = 0;
    try {
        Vector3d origin(-3.0, -3.0, -0.5);
        Matrix3d K;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) K(r, c) = info.K[3 * r + c];
        // a pose a little off the initial one
        Matrix3d R; Vector3d t(0.02, -0.01, 1.03);
        const double rr[9] = {1, 0, 0, 0, 0, -1, 0, -1, 0};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = rr[3 * r + c];

        // 1. field assignment == set_camera_transformation
        SDF a(m, 6.0, 6.0, 3.5, origin, 0.3, 0.025), b(m, 6.0, 6.0, 3.5, origin, 0.3, 0.025);
        CameraTracking ta(20, 0.001, 1.0, 0.01, &a), tb(20, 0.001, 1.0, 0.01, &b);
        ta.set_K(K); tb.K = K; tb.isKFilled = true;            // K assigned, too
        ta.set_camera_transformation(R, t);
        tb.rot = R; tb.trans = t;
        a.update(&ta, clouds[0], nrms[0]);
        b.update(&tb, clouds[0], nrms[0]);
        const bool ok1 = same_volume(&a, &b) && tb.rot_inv(1, 2) == ta.rot_inv(1, 2) && tb.rot_inv_trans(2) == ta.rot_inv_trans(2);
        std::printf("%s field writes reach the integration\n", ok1 ? "ok" : "FAIL"); bad += !ok1;

        // 2. other tracker constants reconfigure the handle
        CameraTracking coarse(5, 0.002f, 0.5f, 0.02f, &b);
        tsdf_config cfg; tsdf_get_config(b.handle(), &cfg);
        const bool ok2 = cfg.gn_max_iter == 5 && cfg.max_twist_diff == 0.002f && cfg.v_h == 0.5f && cfg.w_h == 0.02f;
        std::printf("%s tracker constants taken from the CameraTracking constructor\n", ok2 ? "ok" : "FAIL"); bad += !ok2;

        // 3. a cloud changed in place after tracking is uploaded again
        SDF c(m, 6.0, 6.0, 3.5, origin, 0.3, 0.025), d(m, 6.0, 6.0, 3.5, origin, 0.3, 0.025);
        CameraTracking tc(20, 0.001, 1.0, 0.01, &c), td(20, 0.001, 1.0, 0.01, &d);
        tc.set_K(K); td.set_K(K);
        c.update(&tc, clouds[0], nrms[0]); d.update(&td, clouds[0], nrms[0]);
        tc.estimate_new_position(&c, clouds[1]);
        td.estimate_new_position(&d, clouds[1]);
        Matrix3d Rc = tc.rot; Vector3d tcv = tc.trans;
        td.set_camera_transformation(Rc, tcv);                  // (same pose anyway; keeps the two bit-identical)
        {   // in place, same array, ONE point (and not one a sampled token would look at): every point is compared
            size_t i = clouds[1]->points.size() / 2 + 37;
            while (!(clouds[1]->points[i].z == clouds[1]->points[i].z)) ++i;
            clouds[1]->points[i].z += 0.05f;
        }
        c.update(&tc, clouds[1], nrms[1]);                      // must notice and upload the points again
        CloudPtr fresh(new pcl::PointCloud<pcl::PointXYZRGB>(*clouds[1]));                         // another array: no reuse possible
        d.update(&td, fresh, nrms[1]);
        const bool ok3 = same_volume(&c, &d);
        std::printf("%s a cloud changed in place is uploaded again\n", ok3 ? "ok" : "FAIL"); bad += !ok3;

        // 4. public helpers: values go to stdout for the oracle comparison (tests/test_cpp_shim.py)
        SDF e(m, 6.0, 6.0, 3.5, origin, 0.3, 0.025);
        CameraTracking te(20, 0.001, 1.0, 0.01, &e);
        te.set_K(K);
        e.update(&te, clouds[0], nrms[0]);
        Vector3i ijk; ijk(0) = 3; ijk(1) = m - 1; ijk(2) = 7;
        Vector3i back, outside; outside(0) = 0; outside(1) = m; outside(2) = 0;
        const int idx = e.get_array_index(ijk);
        e.get_voxel_coordinates(idx, back);
        Vector3d g, vox, cam, img3, world;
        Vector2d img;
        e.get_global_coordinates(ijk, g);
        e.get_voxel_coordinates(g, vox);
        te.project_world_to_camera(g, cam);
        te.project_camera_to_image_plane(cam, img);
        te.project_camera_to_world(cam, world);
        std::printf("H idx %d %d %d %d %d\n", idx, e.get_array_index(outside), back(0), back(1), back(2));
        std::printf("H geo %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", g(0), g(1), g(2), vox(0), vox(1), vox(2),
                    cam(0), cam(1), cam(2), img(0), img(1), world(0), world(1), world(2));
        double p2p = 0;
        e.projectivePointToPlaneDistance(cam, g, world, p2p);
        std::printf("H p2p %.17g\n", p2p);
        int n_interp = 0;
        for (int k = 0; k < 40; ++k) {
            const pcl::PointXYZRGB& p = clouds[0]->points[((size_t)(h / 2 - 20 + k) * w + (size_t)(w / 4 + 3 * k))];
            if (!(p.x == p.x)) continue;
            Vector3d cp((double)p.x, (double)p.y, (double)p.z);
            Eigen::Matrix<double, 6, 1> J;
            for (int q = 0; q < 6; ++q) J(q) = -7.0;            // untouched entries stay recognisable
            bool isi = false; double val = -7.0;
            te.get_partial_derivative(&e, cp, J, isi, val);
            n_interp += isi ? 1 : 0;
            std::printf("H J %.9g %.9g %.9g %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (double)p.x, (double)p.y, (double)p.z, isi ? 1 : 0, val,
                        J(0), J(1), J(2), J(3), J(4), J(5));
        }
        const bool ok4 = idx == m * m * 3 + m * (m - 1) + 7 && back(0) == 3 && back(1) == m - 1 && back(2) == 7 && n_interp > 10;
        std::printf("%s public helpers of the two classes\n", ok4 ? "ok" : "FAIL"); bad += !ok4;
    } catch (const tsdf_shim::Error& e) {
        std::fprintf(stderr, "tsdf error %d: %s\n", e.code, e.what());
        return 1;
    }
    return bad ? 1 : 0;
}
