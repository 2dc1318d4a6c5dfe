// Compile-only: the statements of the reference's hot-path call sites, typed as the reference types them, against
// include/sdf_3d_reconstruction/hotpath.hpp with -DTSDF_WITH_EIGEN_PCL -DTSDF_WITH_ROS.  The reference's
// sdf_reconstruction.h would include the shim in place of sdf.h / camera_tracking.h; everything below is then the
// reference's own code shape (sdf_reconstruction.cpp: constructor :83-88 and :90-91, ground-truth branch :65,
// tracking :70-71, fusion :74).  Linking it needs libtsdf_hip.so; the test only compiles it.
#include "sdf_3d_reconstruction/hotpath.hpp"

using namespace Eigen;

struct Node {
    SDF* sdf;
    CameraTracking* camera_tracking;
    bool _useGroundTruth;
    int frame_num;

    void writePoseToFile(double stamp, const Eigen::Vector3d& trans, const Eigen::Matrix3d& rot) {
        (void)stamp; (void)trans.x(); (void)rot(0, 0);
    }

    Node() {
        Vector3d sdf_origin(-3.0, -3.0, -0.5);
        sdf = new SDF(256, 6.0, 6.0, 3.5, sdf_origin, 0.3, 0.025);                       // :85
        _useGroundTruth = false;
        this->camera_tracking = new CameraTracking(20, 0.001, 1.0, 0.01, sdf);           // :88
        frame_num = 0;
    }

    void camera_info(const sensor_msgs::CameraInfoConstPtr& msg) {
        void (CameraTracking::*cb)(const sensor_msgs::CameraInfoConstPtr&) = &CameraTracking::camera_info_cb;   // :90-91
        (this->camera_tracking->*cb)(msg);
        this->camera_tracking->cam_info.shutdown();
        if (!this->camera_tracking->isKFilled) return;
    }

    void kinect_callback(pcl::PointCloud<pcl::PointXYZRGB>::Ptr cloud_filtered, pcl::PointCloud<pcl::Normal>::Ptr normals,
                         double stamp, Matrix3d rotMat, Vector3d trans) {
        frame_num++;
        if (_useGroundTruth) {
            this->camera_tracking->set_camera_transformation(rotMat, trans);             // :65
        } else {
            if (frame_num > 1) {
                this->camera_tracking->estimate_new_position(sdf, cloud_filtered);       // :70
                writePoseToFile(stamp, this->camera_tracking->trans, this->camera_tracking->rot);   // :71
            }
        }
        sdf->update(this->camera_tracking, cloud_filtered, normals);                     // :74
        bool ok = false;
        Vector3d v(1.0, 2.0, 3.0);
        (void)sdf->interpolate_distance(v, ok);                                          // sdf.h:86
        (void)sdf->m; (void)sdf->m_div_width; (void)sdf->get_number_of_voxels();
        (void)this->camera_tracking->rot_inv(0, 0); (void)this->camera_tracking->rot_inv_trans(0); (void)this->camera_tracking->K(2, 2);
    }

    // not in sdf_reconstruction.cpp, but legal against the reference's headers: the pose fields are plain public members
    // (camera_tracking.h:43-59), the tracker takes any constants (camera_tracking.cpp:3-4)
    void other_callers(pcl::PointCloud<pcl::PointXYZRGB>::Ptr cloud_filtered, pcl::PointCloud<pcl::Normal>::Ptr normals,
                       Matrix3d rotMat, Vector3d trans) {
        this->camera_tracking->rot = rotMat;                                             // written through at the next hot call
        this->camera_tracking->trans = trans;
        this->camera_tracking->K(0, 0) = 520.0;
        sdf->update(this->camera_tracking, cloud_filtered, normals);
        CameraTracking* coarse = new CameraTracking(5, 0.002, 0.5, 0.02, sdf);           // other constants than the SDF's defaults
        coarse->estimate_new_position(sdf, cloud_filtered);
        delete coarse;
        // the rest of the public surface (sdf.h:113-181, camera_tracking.h:69-101): any other caller compiles, too
        Vector3i ijk; ijk(0) = 1; ijk(1) = 2; ijk(2) = 3;
        Vector3d g, vox, cam, world;
        Vector2d px;
        const int idx = sdf->get_array_index(ijk);
        sdf->get_voxel_coordinates(idx, ijk);
        sdf->get_global_coordinates(ijk, g);
        sdf->get_voxel_coordinates(g, vox);
        this->camera_tracking->project_world_to_camera(g, cam);
        this->camera_tracking->project_camera_to_image_plane(cam, px);
        this->camera_tracking->project_camera_to_world(cam, world);
        double d = 0;
        sdf->projectivePointToPlaneDistance(cam, world, g, d);
        sdf->projectivePointToPointDistance(cam(2), world(2), d);
        Eigen::Matrix<double, 6, 1> SDF_derivative;
        bool is_interpolated = false;
        double sdf_val = 0;
        this->camera_tracking->get_partial_derivative(sdf, cam, SDF_derivative, is_interpolated, sdf_val);
    }
};

int main() {
    Node n;
    (void)n;
    return 0;
}
