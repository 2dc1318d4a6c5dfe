// hotpath.hpp -- header-only C++ shim that keeps the reference's two hot-path classes, `SDF` and
// `CameraTracking`, with their method names and argument order, on top of the C ABI of tsdf.h.
//
// It is what a maintainer of mees/tracking_sdf would include from sdf_reconstruction.cpp instead of
// sdf.h / camera_tracking.h to run the per-frame hot path on an MI355X.  Eigen and
// PCL are not required: matrices are plain row-major arrays (`Mat3`, `Vec3`), clouds are the small
// `OrganizedCloud` / `NormalCloud` views below.  With -DTSDF_WITH_EIGEN_PCL (and those headers on the
// include path) the block at the end of this file adds `SDF` / `CameraTracking` classes with the reference's exact
// signatures (Eigen::Vector3d& origin, pcl::PointCloud<...>::Ptr clouds, Eigen-typed public rot / trans / K) at
// global scope, so that the reference's call sites compile unchanged (tests/test_cpp_shim.py compiles them against
// minimal mock headers).
//
// Reference interfaces mirrored (paths relative to the reference's src/):
//   SDF::SDF(m, width, height, depth, origin, delta, epsilon)           include/sdf_3d_reconstruction/sdf.h:78-79
//   SDF::update(CameraTracking*, cloud_filtered, normals)               sdf.h:161-163
//   SDF::interpolate_distance(voxel_coordinates, is_interpolated)       sdf.h:86
//   SDF::mesh(vertices, colors)          the body of SDF::visualize       sdf.cpp:317-391 (mc->performReconstruction + interpolate_color)
//   SDF::m, m_div_width/height/depth, get_number_of_voxels()            sdf.h:69-72,107
//   CameraTracking::CameraTracking(max_iter, max_twist_diff, v_h, w_h, sdf)   camera_tracking.cpp:3-4 (definition order)
//   CameraTracking::estimate_new_position(sdf, point_cloud)             camera_tracking.h:101
//   CameraTracking::set_camera_transformation(rot, trans)               camera_tracking.h:84
//   CameraTracking::camera_info_cb -> set_K(K)                          camera_tracking.cpp:22-36
//   public rot, trans, rot_inv, rot_inv_trans, K, isKFilled             camera_tracking.h:43-59
// and, in the exact-type classes only (host-side one-liners over the pose / configuration of the handle, with the
// reference's arithmetic and evaluation order):
//   SDF::get_array_index / get_voxel_coordinates (both) / get_global_coordinates / projectivePoint*Distance   sdf.h:113-181
//   CameraTracking::project_camera_to_image_plane / project_world_to_camera / project_camera_to_world   camera_tracking.cpp:40-58
//   CameraTracking::get_partial_derivative (13 look-ups in ONE tsdf_sample call)                       camera_tracking.cpp:246-363
// Not mirrored: SDF::visualize (the 1 Hz ROS marker thread, sdf.cpp:317-391; its mesh is SDF::mesh), the analytic
// fillers create_circle / create_cuboid (no callers in the reference).
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../tsdf.h"

namespace tsdf_shim {

using Vec3 = std::array<double, 3>;
using Mat3 = std::array<double, 9>;   // row-major

// Non-owning views of the two inputs of the hot calls (PCL's organised clouds: at(col,row) = [row*width+col]).
struct OrganizedCloud {
    const float* xyz = nullptr;      // height*width*3, NaN = no depth
    const uint8_t* rgb = nullptr;    // height*width*3 (r,g,b), may be null
    int32_t width = 0, height = 0;
    OrganizedCloud() {}
    OrganizedCloud(const float* xyz_, const uint8_t* rgb_, int32_t w, int32_t h) : xyz(xyz_), rgb(rgb_), width(w), height(h) {}
};
struct NormalCloud {
    const float* normal = nullptr;   // height*width*3, NaN = undefined
    int32_t width = 0, height = 0;
    NormalCloud() {}
    NormalCloud(const float* n, int32_t w, int32_t h) : normal(n), width(w), height(h) {}
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class CameraTracking;

class SDF {
public:
    int m;
    float m_div_height, m_div_width, m_div_depth;

    // standard constructor (sdf.h:78-79); `base` lets the caller override placement / colour / tracker steps
    SDF(int m_, float width, float height, float depth, const Vec3& sdf_origin, float distance_delta,
        float distance_epsilon, const tsdf_config* base = nullptr)
        : m(m_), m_div_height(m_ / height), m_div_width(m_ / width), m_div_depth(m_ / depth) {
        if (tsdf_abi_version() != TSDF_ABI_VERSION)       // struct layouts differ between versions
            throw Error(TSDF_E_BADARG, "libtsdf_hip.so has ABI version " + std::to_string(tsdf_abi_version()) +
                                       ", this header is version " + std::to_string(TSDF_ABI_VERSION) + ": rebuild");
        tsdf_config cfg;
        if (base) cfg = *base; else tsdf_default_config(&cfg);
        cfg.m = m_; cfg.width = width; cfg.height = height; cfg.depth = depth;
        cfg.origin[0] = sdf_origin[0]; cfg.origin[1] = sdf_origin[1]; cfg.origin[2] = sdf_origin[2];
        cfg.delta = distance_delta; cfg.epsilon = distance_epsilon;
        const int rc = tsdf_create(&cfg, &h_);
        if (rc != TSDF_OK) throw Error(rc, std::string("tsdf_create: ") + tsdf_last_error(nullptr));
    }
    ~SDF() { tsdf_destroy(h_); }
    SDF(const SDF&) = delete;
    SDF& operator=(const SDF&) = delete;

    int get_number_of_voxels() const { return m * m * m; }

    // sdf.h:161-163.  The tracker argument is kept for signature compatibility: pose and intrinsics live
    // in the same native handle.
    inline void update(CameraTracking* camera_tracking, const OrganizedCloud& cloud_filtered, const NormalCloud& normals);

    // sdf.h:86
    float interpolate_distance(const Vec3& voxel_coordinates, bool& is_interpolated) const {
        float val = 0.f;
        int32_t ok = 0;
        check(tsdf_sample(h_, voxel_coordinates.data(), 1, &val, &ok), "tsdf_sample");
        is_interpolated = ok != 0;
        return val;
    }

    // Raw depth image instead of PCL clouds: back-projection, bilateral filter and normals on the GPU
    // (tsdf_set_depth_frame; stand-in for the PCL calls of sdf_reconstruction.cpp:37-49).  The frame stays
    // current: follow with CameraTracking::estimate_new_position(sdf) and SDF::update(tracker).
    void set_depth_frame(const uint16_t* depth16, const uint8_t* rgb, int32_t width, int32_t height,
                         const tsdf_preproc_params* params = nullptr) {
        check(tsdf_set_depth_frame(h_, depth16, nullptr, rgb, width, height, params), "tsdf_set_depth_frame");
    }
    inline void update(CameraTracking* camera_tracking);   // integrate the current frame

    // host mirrors of D / W for a mesher (the reference hands raw pointers to MarchingCubesSDF, sdf.cpp:47-49)
    void download(std::vector<float>& D, std::vector<float>& W) const {
        tsdf_config c;
        tsdf_get_config(h_, &c);
        const size_t n = (size_t)(c.slab_x1 - c.slab_x0) * m * m;
        D.resize(n); W.resize(n);
        check(tsdf_download(h_, D.data(), W.data()), "tsdf_download");
    }

    // What the visualiser thread computes every tick (SDF::visualize, sdf.cpp:317-391): marching cubes over the
    // observed cubes + (optionally) SDF::interpolate_color at every vertex, on the GPU.  vertices: 9 floats per
    // triangle in the grid-local frame of pcl::MarchingCubesSDF::performReconstruction (add sdf_origin for the
    // marker points, sdf.cpp:355-369); colors: 4 floats per vertex.  Returns the number of triangles.
    int64_t mesh(std::vector<float>& vertices, std::vector<float>* colors = nullptr, float iso_level = 0.0f) const {
        int64_t n = 0;
        check(tsdf_mesh_extract(h_, iso_level, colors ? 1 : 0, &n), "tsdf_mesh_extract");
        vertices.resize((size_t)n * 9);
        if (colors) colors->resize((size_t)n * 12);
        check(tsdf_mesh_read(h_, vertices.data(), colors ? colors->data() : nullptr, n), "tsdf_mesh_read");
        return n;
    }

    tsdf_handle* handle() const { return h_; }
    void check(int rc, const char* what) const {
        if (rc != TSDF_OK) throw Error(rc, std::string(what) + ": " + tsdf_last_error(h_));
    }

private:
    tsdf_handle* h_ = nullptr;
};

class CameraTracking {
public:
    Mat3 rot{}, rot_inv{}, K{};
    Vec3 trans{}, rot_inv_trans{};
    bool isKFilled = false;

    // definition order of the reference: (gauss_newton_max_iteration, maximum_twist_diff, v_h, w_h, sdf).  The reference
    // takes any constants here (camera_tracking.cpp:3-18); the native handle got its own when the SDF was created, so
    // the ones given here reconfigure it.
    CameraTracking(int gauss_newton_max_iteration, float maximum_twist_diff, float v_h, float w_h, SDF* sdf)
        : sdf_(sdf) {
        sdf->check(tsdf_set_tracker_params(sdf->handle(), gauss_newton_max_iteration, maximum_twist_diff, v_h, w_h),
                   "tsdf_set_tracker_params");
        sync();
    }

    void set_K(const Mat3& k) {              // camera_info_cb, camera_tracking.cpp:22-36
        K = k;
        sdf_->check(tsdf_set_intrinsics(sdf_->handle(), K.data()), "tsdf_set_intrinsics");
        isKFilled = true;
    }
    void set_camera_transformation(const Mat3& r, const Vec3& t) {   // camera_tracking.cpp:59-65
        sdf_->check(tsdf_set_camera_transformation(sdf_->handle(), r.data(), t.data()), "tsdf_set_camera_transformation");
        sync();
    }
    // camera_tracking.h:101; on a singular system / no samples the pose is left unchanged and Error is thrown
    // (the reference silently continues with a NaN pose, camera_tracking.cpp:191)
    void estimate_new_position(const SDF* sdf, const OrganizedCloud& point_cloud, tsdf_track_stats* stats = nullptr) {
        sdf->check(tsdf_set_frame(sdf->handle(), point_cloud.xyz, nullptr, point_cloud.rgb, point_cloud.width,
                                  point_cloud.height), "tsdf_set_frame");
        const int rc = tsdf_track(sdf->handle(), stats);
        sync();
        sdf->check(rc, "tsdf_track");
    }
    // same, against the frame that is already current (after SDF::set_depth_frame)
    void estimate_new_position(const SDF* sdf, tsdf_track_stats* stats = nullptr) {
        const int rc = tsdf_track(sdf->handle(), stats);
        sync();
        sdf->check(rc, "tsdf_track");
    }
    // estimate_new_position(sdf) + sdf->update(this) in one call (the pair of sdf_reconstruction.cpp:70,74): the
    // integration is launched the moment the last Gauss-Newton pass is solved
    void estimate_new_position_and_update(const SDF* sdf, tsdf_track_stats* stats = nullptr) {
        const int rc = tsdf_track_and_integrate(sdf->handle(), 1, stats, nullptr);
        sync();
        sdf->check(rc, "tsdf_track_and_integrate");
    }
    void sync() {
        tsdf_get_pose(sdf_->handle(), rot.data(), trans.data(), rot_inv.data(), rot_inv_trans.data());
    }

private:
    SDF* sdf_;
};

inline void SDF::update(CameraTracking* camera_tracking, const OrganizedCloud& cloud_filtered, const NormalCloud& normals) {
    (void)camera_tracking;
    check(tsdf_set_frame(h_, cloud_filtered.xyz, normals.normal, cloud_filtered.rgb, cloud_filtered.width,
                         cloud_filtered.height), "tsdf_set_frame");
    check(tsdf_integrate(h_, nullptr), "tsdf_integrate");      // reference: exit(0) when K is missing (sdf.cpp:227-230)
}

inline void SDF::update(CameraTracking* camera_tracking) {
    (void)camera_tracking;
    check(tsdf_integrate(h_, nullptr), "tsdf_integrate");
}

}  // namespace tsdf_shim

#ifdef TSDF_WITH_EIGEN_PCL
// The reference's exact argument types.  Compiled only where Eigen and PCL exist (they do not in the build
// container; tests/mock/ holds minimal stand-ins of the few declarations used, for a compile-only test).
// `SDF` and `CameraTracking` below have the reference's constructor / method signatures and Eigen-typed public
// fields, so that the hot-path call sites of sdf_reconstruction.cpp (:70 estimate_new_position, :71 writePoseToFile
// reading trans / rot, :65 set_camera_transformation, :74 update, :83-88 the two constructors) compile unchanged
// once sdf_reconstruction.h includes this header instead of sdf.h / camera_tracking.h.  PCL's 32-byte AoS points go
// to the library as they are (tsdf_set_frame_aos repacks them into its pinned staging planes with a few threads).  With TSDF_WITH_ROS also camera_info_cb and the cam_info subscriber
// (camera_tracking.cpp:22-36, sdf_reconstruction.cpp:90-91).
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#ifdef TSDF_WITH_ROS
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#endif
namespace tsdf_shim {
// PCL clouds -> the planes of the plain-type classes above (a host-side repack; the exact-type classes below hand the
// clouds to tsdf_set_frame_aos instead and do not use it)
struct PclFrame {
    std::vector<float> xyz, nrm;
    std::vector<uint8_t> rgb;
    OrganizedCloud cloud;
    NormalCloud normals;
    PclFrame(const pcl::PointCloud<pcl::PointXYZRGB>& c, const pcl::PointCloud<pcl::Normal>* n) {
        const size_t np = (size_t)c.width * c.height;
        xyz.resize(np * 3); rgb.resize(np * 3);
        for (size_t i = 0; i < np; ++i) {
            const pcl::PointXYZRGB& p = c.points[i];
            xyz[3 * i] = p.x; xyz[3 * i + 1] = p.y; xyz[3 * i + 2] = p.z;
            rgb[3 * i] = p.r; rgb[3 * i + 1] = p.g; rgb[3 * i + 2] = p.b;
        }
        cloud.xyz = xyz.data(); cloud.rgb = rgb.data(); cloud.width = (int32_t)c.width; cloud.height = (int32_t)c.height;
        if (n) {
            nrm.resize(np * 3);
            for (size_t i = 0; i < np; ++i) {
                nrm[3 * i] = n->points[i].normal_x; nrm[3 * i + 1] = n->points[i].normal_y; nrm[3 * i + 2] = n->points[i].normal_z;
            }
            normals.normal = nrm.data(); normals.width = (int32_t)c.width; normals.height = (int32_t)c.height;
        }
    }
};

// Byte layout of the two PCL point types as this build sees them (PCL pads both to 32 bytes; measured on an instance,
// offsetof is not defined for them), for tsdf_set_frame_aos: the clouds go to the library as they are.
inline const tsdf_aos_layout& pcl_layout() {
    static const tsdf_aos_layout lay = [] {
        const pcl::PointXYZRGB p = pcl::PointXYZRGB();
        const pcl::Normal n = pcl::Normal();
        const char* pb = reinterpret_cast<const char*>(&p);
        const char* nb = reinterpret_cast<const char*>(&n);
        tsdf_aos_layout l;
        l.point_stride = (int32_t)sizeof(pcl::PointXYZRGB);
        l.xyz_offset = (int32_t)(reinterpret_cast<const char*>(&p.x) - pb);
        l.r_offset = (int32_t)(reinterpret_cast<const char*>(&p.r) - pb);
        l.g_offset = (int32_t)(reinterpret_cast<const char*>(&p.g) - pb);
        l.b_offset = (int32_t)(reinterpret_cast<const char*>(&p.b) - pb);
        l.normal_stride = (int32_t)sizeof(pcl::Normal);
        l.normal_offset = (int32_t)(reinterpret_cast<const char*>(&n.normal_x) - nb);
        return l;
    }();
    return lay;
}

namespace ref_types {

inline Mat3 to_rows(const Eigen::Matrix3d& M) {
    Mat3 a;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a[3 * r + c] = M(r, c);
    return a;
}
inline void from_rows(const Mat3& a, Eigen::Matrix3d& M) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M(r, c) = a[3 * r + c];
}

class CameraTracking;

class SDF : public tsdf_shim::SDF {
public:
    // sdf.h:78-79
    SDF(int m_, float width, float height, float depth, Eigen::Vector3d& sdf_origin, float distance_delta,
        float distance_epsilon, const tsdf_config* base = nullptr)
        : tsdf_shim::SDF(m_, width, height, depth, Vec3{{sdf_origin(0), sdf_origin(1), sdf_origin(2)}}, distance_delta,
                         distance_epsilon, base) {}
    // sdf.h:161-163
    inline void update(CameraTracking* camera_tracking, pcl::PointCloud<pcl::PointXYZRGB>::Ptr cloud_filtered,
                       pcl::PointCloud<pcl::Normal>::Ptr normals);
    // sdf.h:86
    float interpolate_distance(const Eigen::Vector3d& voxel_coordinates, bool& is_interpolated) const {
        return tsdf_shim::SDF::interpolate_distance(Vec3{{voxel_coordinates(0), voxel_coordinates(1), voxel_coordinates(2)}},
                                                    is_interpolated);
    }
    using tsdf_shim::SDF::update;
    using tsdf_shim::SDF::interpolate_distance;

    // sdf.h:113-127: (i,j,k) -> array index, -1 outside the grid.  The reference's `int` index is kept (and with it its
    // wrap at m >= 1291): callers of this helper index host arrays of the reference's size.
    int get_array_index(Eigen::Vector3i& voxel_coordinates) const {
        if (voxel_coordinates(0) < 0 || voxel_coordinates(1) < 0 || voxel_coordinates(2) < 0) return -1;
        if (voxel_coordinates(0) >= m || voxel_coordinates(1) >= m || voxel_coordinates(2) >= m) return -1;
        const int idx = m * m * voxel_coordinates(0) + m * voxel_coordinates(1) + voxel_coordinates(2);
        return (idx < 0 || idx >= get_number_of_voxels()) ? -1 : idx;
    }
    // sdf.h:132-136
    void get_voxel_coordinates(int array_idx, Eigen::Vector3i& voxel_coordinates) const {
        voxel_coordinates(1) = (int)(array_idx % (m * m)) / m;
        voxel_coordinates(0) = (int)(array_idx / (m * m));
        voxel_coordinates(2) = (int)array_idx % m;
    }
    // sdf.h:143-147: float m_div_* widened against double coordinates
    void get_voxel_coordinates(Eigen::Vector3d& global_coordinates, Eigen::Vector3d& voxel_coordinates) const {
        const tsdf_config c = config();
        voxel_coordinates(0) = ((global_coordinates(0) - c.origin[0]) * m_div_width - 0.5);
        voxel_coordinates(1) = ((global_coordinates(1) - c.origin[1]) * m_div_height - 0.5);
        voxel_coordinates(2) = ((global_coordinates(2) - c.origin[2]) * m_div_depth - 0.5);
    }
    // sdf.h:153-157: (extent / (float)m) is a float quotient
    void get_global_coordinates(Eigen::Vector3i& voxel_coordinates, Eigen::Vector3d& global_coordinates) const {
        const tsdf_config c = config();
        global_coordinates(0) = (c.width / ((float)m)) * (voxel_coordinates(0) + 0.5) + c.origin[0];
        global_coordinates(1) = (c.height / ((float)m)) * (voxel_coordinates(1) + 0.5) + c.origin[1];
        global_coordinates(2) = (c.depth / ((float)m)) * (voxel_coordinates(2) + 0.5) + c.origin[2];
    }
    // sdf.h:169-181
    void projectivePointToPointDistance(const double& voxelDepthInCameraFrame, const double& observedDepthOfProjectedVoxelInDepthImage,
                                        double& pointToPointDistance) const {
        pointToPointDistance = voxelDepthInCameraFrame - observedDepthOfProjectedVoxelInDepthImage;
    }
    void projectivePointToPlaneDistance(const Eigen::Vector3d& camera_point, const Eigen::Vector3d& camera_point_img,
                                        const Eigen::Vector3d& normal, double& pointToPlaneDistance) const {
        const double d0 = camera_point_img(0) - camera_point(0), d1 = camera_point_img(1) - camera_point(1),
                     d2 = camera_point_img(2) - camera_point(2);
        pointToPlaneDistance = d0 * normal(0) + (d1 * normal(1) + d2 * normal(2));        // Vector3d::dot (redux order)
    }
    tsdf_config config() const { tsdf_config c; tsdf_get_config(handle(), &c); return c; }
};

class CameraTracking {
public:
    Eigen::Matrix3d rot;                 // camera_tracking.h:43-59
    Eigen::Matrix3d rot_inv;
    Eigen::Vector3d trans;
    Eigen::Vector3d rot_inv_trans;
    Eigen::Matrix3d K;
    bool isKFilled;
#ifdef TSDF_WITH_ROS
    ros::Subscriber cam_info;
    void camera_info_cb(const sensor_msgs::CameraInfoConstPtr& rgbd_camera_info) {       // camera_tracking.cpp:22-36
        Eigen::Matrix3d k;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) k(r, c) = rgbd_camera_info->K[3 * r + c];
        set_K(k);
        this->cam_info.shutdown();
    }
#endif
    // camera_tracking.cpp:3-4 (the definition's order: max_iter, max_twist_diff, v_h, w_h -- the declaration in
    // camera_tracking.h:63 swaps the names of the last two floats, the call site passes 1.0, 0.01)
    CameraTracking(int gauss_newton_max_iteration, float maximum_twist_diff, float v_h, float w_h, SDF* sdf)
        : isKFilled(false), impl_(gauss_newton_max_iteration, maximum_twist_diff, v_h, w_h, sdf) { pull(); }
    virtual ~CameraTracking() {}

    void set_K(const Eigen::Matrix3d& k) { impl_.set_K(to_rows(k)); K = k; seen_K_ = k; isKFilled = true; }
    // camera_tracking.h:84
    void set_camera_transformation(Eigen::Matrix3d& r, Eigen::Vector3d& t) {
        impl_.set_camera_transformation(to_rows(r), Vec3{{t(0), t(1), t(2)}});
        pull();
    }
    // camera_tracking.h:101.  tsdf_track_aos: the tracker's 34 240 samples go up first and the Gauss-Newton passes run
    // while the library threads stage the rest of the cloud; the cloud is the caller's again when this returns.
    void estimate_new_position(const SDF* sdf, const pcl::PointCloud<pcl::PointXYZRGB>::Ptr& point_cloud) {
        push(sdf);
        const int rc = tsdf_track_aos(sdf->handle(), point_cloud->points.data(), &pcl_layout(), (int32_t)point_cloud->width,
                                      (int32_t)point_cloud->height, nullptr);
        pull();
        sdf->check(rc, "tsdf_track_aos");
    }
    // NOT in the reference: the same call with the frame's normals, which kinect_callback has in hand before it tracks
    // (sdf_reconstruction.cpp:44-49).  One more argument at :70 -- estimate_new_position(sdf, cloud_filtered, normals) --
    // lets the library stage the WHOLE frame under the Gauss-Newton passes; update(tracker, cloud_filtered, normals) with
    // the same two clouds then integrates what was staged without another upload or comparison: the caller vouches that
    // neither cloud changes between the two calls (pass other clouds, or use the two-argument form, when that is not so).
    void estimate_new_position(const SDF* sdf, const pcl::PointCloud<pcl::PointXYZRGB>::Ptr& point_cloud,
                               const pcl::PointCloud<pcl::Normal>::Ptr& normals) {
        push(sdf);
        vouched_points_ = vouched_normals_ = nullptr;
        const int rc = tsdf_track_frame_aos(sdf->handle(), point_cloud->points.data(), normals->points.data(), &pcl_layout(),
                                            (int32_t)point_cloud->width, (int32_t)point_cloud->height, nullptr);
        pull();
        sdf->check(rc, "tsdf_track_frame_aos");
        vouched_points_ = point_cloud->points.data(); vouched_normals_ = normals->points.data();
        vouched_serial_ = tsdf_frame_serial(sdf->handle());
    }
    // one-shot: does update() receive the very clouds the three-argument estimate_new_position staged?
    bool take_vouched(const SDF* sdf, const void* points, const void* normals) {
        const bool yes = vouched_points_ && points == vouched_points_ && normals == vouched_normals_ &&
                         vouched_serial_ == tsdf_frame_serial(sdf->handle());
        vouched_points_ = vouched_normals_ = nullptr;
        return yes;
    }
    // camera_tracking.cpp:40-47: ij = K * camera_point, (u, v) = ij.xy / ij.z   (fixed-size product, Eigen 3.2 order)
    void project_camera_to_image_plane(Eigen::Vector3d& camera_point, Eigen::Vector2d& image_point) {
        double ij[3];
        for (int r = 0; r < 3; ++r) ij[r] = (K(r, 0) * camera_point(0) + K(r, 1) * camera_point(1)) + K(r, 2) * camera_point(2);
        image_point(0) = ij[0] / ij[2];
        image_point(1) = ij[1] / ij[2];
    }
    // camera_tracking.cpp:51-54
    void project_world_to_camera(Eigen::Vector3d& world_point, Eigen::Vector3d& camera_point) {
        double o[3];
        for (int r = 0; r < 3; ++r)
            o[r] = ((rot_inv(r, 0) * world_point(0) + rot_inv(r, 1) * world_point(1)) + rot_inv(r, 2) * world_point(2)) + rot_inv_trans(r);
        for (int r = 0; r < 3; ++r) camera_point(r) = o[r];
    }
    // camera_tracking.cpp:55-58
    void project_camera_to_world(const Eigen::Vector3d& camera_point, Eigen::Vector3d& world_point) {
        double o[3];
        for (int r = 0; r < 3; ++r)
            o[r] = ((rot(r, 0) * camera_point(0) + rot(r, 1) * camera_point(1)) + rot(r, 2) * camera_point(2)) + trans(r);
        for (int r = 0; r < 3; ++r) world_point(r) = o[r];
    }
    // camera_tracking.cpp:246-363: data association + numeric Jacobian of ONE camera-frame point against the volume in
    // HBM.  The reference's 13 sequential SDF::interpolate_distance calls become one tsdf_sample call of 13 voxel
    // positions; the outputs are then written in the reference's order and with its early returns: out of the grid ->
    // nothing is touched (:261-268); a look-up without a valid corner -> is_interpolated = false and what was written
    // before it stays (:269-361).  For tools and tests: the tracker itself never comes through here.
    void get_partial_derivative(const SDF* sdf, const Eigen::Vector3d& camera_point, Eigen::Matrix<double, 6, 1>& SDF_derivative,
                                bool& is_interpolated, double& sdf_val) {
        push(sdf);
        const tsdf_config c = sdf->config();
        const float mdiv[3] = {sdf->m_div_width, sdf->m_div_height, sdf->m_div_depth};
        auto voxel_of = [&](const double R[9], double v[3]) {           // project_camera_to_world + get_voxel_coordinates
            for (int r = 0; r < 3; ++r) {
                const double w = ((R[3 * r] * camera_point(0) + R[3 * r + 1] * camera_point(1)) + R[3 * r + 2] * camera_point(2)) + trans(r);
                v[r] = ((w - c.origin[r]) * mdiv[r] - 0.5);
            }
        };
        double R0[9];
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) R0[3 * r + q] = rot(r, q);
        double vox[13][3];
        voxel_of(R0, vox[0]);
        for (int a = 0; a < 3; ++a) {
            if (vox[0][a] < 0) return;                                  // :261-264
        }
        for (int a = 0; a < 3; ++a) {
            if (vox[0][a] >= sdf->m) return;                            // :265-268
        }
        for (int a = 0; a < 3; ++a) {                                   // :273-316, voxel space
            for (int q = 0; q < 3; ++q) { vox[1 + 2 * a][q] = vox[0][q]; vox[2 + 2 * a][q] = vox[0][q]; }
            vox[1 + 2 * a][a] += c.v_h;
            vox[2 + 2 * a][a] -= c.v_h;
        }
        for (int a = 0; a < 3; ++a) {                                   // :92-145 + :318-361: (I +- w_h [e_a]x) rot, about the world axes
            for (int sgn = 0; sgn < 2; ++sgn) {
                const double w = sgn == 0 ? (double)c.w_h : -(double)c.w_h;
                double Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Rp[9];
                const int b1 = (a + 1) % 3, b2 = (a + 2) % 3;
                Rd[3 * b1 + b2] = -w; Rd[3 * b2 + b1] = w;
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q)
                        Rp[3 * r + q] = (Rd[3 * r] * R0[q] + Rd[3 * r + 1] * R0[3 + q]) + Rd[3 * r + 2] * R0[6 + q];
                voxel_of(Rp, vox[7 + 2 * a + sgn]);
            }
        }
        float val[13];
        int32_t ok[13];
        sdf->check(tsdf_sample(sdf->handle(), &vox[0][0], 13, val, ok), "tsdf_sample");
        sdf_val = val[0];                                               // :269
        is_interpolated = ok[0] != 0;
        if (!is_interpolated) return;
        const float v_h2 = 2 * c.v_h;
        const float vh2[3] = {v_h2 / sdf->m_div_width, v_h2 / sdf->m_div_height, v_h2 / sdf->m_div_depth};   // camera_tracking.cpp:13-17
        for (int e = 0; e < 6; ++e) {
            const int ip = e < 3 ? 1 + 2 * e : 7 + 2 * (e - 3), im = ip + 1;
            is_interpolated = ok[ip] != 0;
            if (!is_interpolated) return;
            is_interpolated = ok[im] != 0;
            if (!is_interpolated) return;
            SDF_derivative(e) = e < 3 ? (val[ip] - val[im]) / vh2[e] : (val[ip] - val[im]) / (2 * c.w_h);     // float quotients widened
        }
    }
    tsdf_shim::CameraTracking* impl() { return &impl_; }
    void pull() {                          // native handle -> the public Eigen fields
        impl_.sync();
        from_rows(impl_.rot, rot); from_rows(impl_.rot_inv, rot_inv);
        for (int a = 0; a < 3; ++a) { trans(a) = impl_.trans[a]; rot_inv_trans(a) = impl_.rot_inv_trans[a]; }
        seen_rot_ = rot; seen_trans_ = trans; seen_rot_inv_ = rot_inv; seen_rot_inv_trans_ = rot_inv_trans;
    }
    // The public fields are plain members in the reference (camera_tracking.h:43-59) and the tracker / SDF::update read
    // them directly, so a caller may simply assign them.  Here they mirror the native handle: before every hot call
    // a pose or K that no longer equals what was last pulled is written through (rot / trans by
    // set_camera_transformation, which also renews rot_inv / rot_inv_trans as camera_tracking.cpp:59-65 does).
    // Assigning rot_inv / rot_inv_trans alone -- a pose and an inverse that do not belong together -- has no
    // counterpart in the native handle and is refused.
    void push(const SDF* sdf) {
        if (isKFilled && !(K == seen_K_)) set_K(Eigen::Matrix3d(K));
        if (!(rot == seen_rot_) || !(trans == seen_trans_)) {
            Eigen::Matrix3d r = rot; Eigen::Vector3d t = trans;
            set_camera_transformation(r, t);
        } else if (!(rot_inv == seen_rot_inv_) || !(rot_inv_trans == seen_rot_inv_trans_)) {
            throw Error(TSDF_E_BADARG, "CameraTracking: rot_inv / rot_inv_trans were assigned without rot / trans");
        }
        (void)sdf;
    }
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW

private:
    tsdf_shim::CameraTracking impl_;
    const void* vouched_points_ = nullptr; const void* vouched_normals_ = nullptr; int64_t vouched_serial_ = -1;
    Eigen::Matrix3d seen_rot_, seen_rot_inv_, seen_K_;
    Eigen::Vector3d seen_trans_, seen_rot_inv_trans_;
};

// sdf.h:161-163.  The reference hands the cloud it has just tracked to SDF::update (sdf_reconstruction.cpp:70,74): its
// points are in HBM already and tsdf_integrate_aos uploads the normals only -- after the LIBRARY has checked, on its
// staging threads and under the normals' copy, that the cloud still holds the bytes that were tracked (every point: a
// cloud filtered in place between the two calls, or another one at the same address, is uploaded again and integrated as
// it is now, as sdf.cpp:258-259 would read it).
inline void SDF::update(CameraTracking* camera_tracking, pcl::PointCloud<pcl::PointXYZRGB>::Ptr cloud_filtered,
                        pcl::PointCloud<pcl::Normal>::Ptr normals) {
    if (camera_tracking) camera_tracking->push(this);
    if (camera_tracking && camera_tracking->take_vouched(this, cloud_filtered->points.data(), normals->points.data())) {
        check(tsdf_integrate(handle(), nullptr), "tsdf_integrate");      // staged whole by the three-argument estimate_new_position
        return;
    }
    check(tsdf_integrate_aos(handle(), cloud_filtered->points.data(), normals->points.data(), &pcl_layout(),
                             (int32_t)cloud_filtered->width, (int32_t)cloud_filtered->height, nullptr), "tsdf_integrate_aos");
}

}  // namespace ref_types
}  // namespace tsdf_shim

#ifndef TSDF_SHIM_NO_GLOBAL_NAMES
// the reference declares both classes at global scope (sdf.h:35, camera_tracking.h:12)
using tsdf_shim::ref_types::SDF;
using tsdf_shim::ref_types::CameraTracking;
#endif
#endif  // TSDF_WITH_EIGEN_PCL
