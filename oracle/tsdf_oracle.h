/*
 * tsdf_oracle.h -- CPU restatement ("oracle") of tracking_sdf's per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tracking_sdf_amd/ or include/ may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / baseline.
 *
 * PARITY UNPINNED vs the real reference binary: the reference
 * (mees/tracking_sdf) ships no tests and no golden vectors, and it cannot be
 * compiled here (every hot-path translation unit pulls in ROS, PCL, Eigen and
 * boost, none of which exist in this image).  The restatement is therefore
 * pinned only by known-answer tests derived by hand from the cited reference
 * lines (tests/test_oracle_kat.py) and by a
 * second restatement in NumPy (oracle/np_oracle.py: written from the reference's
 * lines, whole-array operations instead of these loops; tests/test_np_oracle.py
 * and tools/make_golden.py require the two to agree bit for bit on volume,
 * interpolation and normal equations, to 1e-11 on the tracked pose).
 *
 * Eigen evaluation orders restated by hand (Eigen 3.2.x, the version of the
 * reference's Ubuntu 12.04/14.04 era; the reference does not pin one):
 *   - fixed 3x3 * 3-vector and 3x3 * 3x3 products: coefficient based,
 *     sequential  ((a0*b0 + a1*b1) + a2*b2)
 *   - Vector3d::dot / norm / 3-term sum():  redux unroller  a0*b0 + (a1*b1 + a2*b2)
 *   - Matrix3d::inverse(): cofactor formula, det = c0*m00 + (c1*m10 + c2*m20)
 *   - 6x6 inverse(): partial-pivot LU, inverse = solve(I), then inverse * b
 *
 * All citations are relative to /root/reference/src/.
 */
#ifndef TSDF_ORACLE_H_
#define TSDF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Which Eigen the restated reference is "built" with: 32 (default) = Eigen 3.2's sequential fixed-size products
 * ((a0*b0 + a1*b1) + a2*b2); 33 = the lazy-product redux of Eigen >= 3.3, a0*b0 + (a1*b1 + a2*b2).  Process-wide. */
void orc_set_eigen_order(int32_t version);
int32_t orc_get_eigen_order(void);

/* ---- voxel grid (class SDF, include/sdf_3d_reconstruction/sdf.h:35-186) ---- */
typedef struct orc_sdf {
    int32_t m;                       /* sdf.h:69 */
    float width, height, depth;      /* sdf.h:39 */
    float distance_delta, distance_epsilon;
    double sdf_origin[3];            /* sdf.h:61 */
    float m_div_width, m_div_height, m_div_depth; /* sdf.h:70-72, float on purpose */
    int32_t m_squared;               /* sdf.h:66 */
    int64_t number_of_voxels;        /* sdf.h:56 is int; widened so m>1290 does not wrap */
    float *D, *W, *Color_W, *R, *G, *B;   /* sdf.h:41-55 */
    double *global_coords;           /* sdf.h:45  Vector3d[m^3] = 24 B/voxel */
    uint8_t *exp_mask;               /* NOT in the reference: a checker's annotation (orc_sdf_track_exp_band), NULL unless asked
                                        for -- 1 for every voxel whose weight went through exp() (sdf.cpp:277-279) at least once,
                                        i.e. the voxels in which another correctly working exp() may legitimately differ by an ulp */
} orc_sdf;

/* ---- camera tracker state (class CameraTracking, camera_tracking.h:12-105) ---- */
typedef struct orc_tracker {
    double rot[9], trans[3];         /* camera -> world, row-major */
    double rot_inv[9], rot_inv_trans[3];
    double K[9];
    int32_t isKFilled;
    int32_t gauss_newton_max_iteration;
    float maximum_twist_diff;
    float v_h, w_h, v_h2, w_h2;
    float v_h2_width, v_h2_height, v_h2_depth;
} orc_tracker;

/* ---- organised cloud + normals, laid out like the PCL types the reference
 *      receives: 32-byte PointXYZRGB / 32-byte Normal, at(col,row) = [row*width+col] ---- */
typedef struct orc_point { float x, y, z, pad0; uint8_t b, g, r, a; float pad1[3]; } orc_point;
typedef struct orc_normal { float nx, ny, nz, pad0; float curvature; float pad1[3]; } orc_normal;
typedef struct orc_cloud {
    int32_t width, height;
    orc_point *points;
    orc_normal *normals;
} orc_cloud;

typedef struct orc_accum_stats {
    int64_t n_samples;   /* sampled pixels visited                                  */
    int64_t n_nan;       /* skipped: NaN xyz              camera_tracking.cpp:168   */
    int64_t n_oog;       /* centre voxel out of grid       camera_tracking.cpp:261-268 */
    int64_t n_fail;      /* in grid, some look-up had no valid corner               */
    int64_t n_ok;        /* in grid, all 13 look-ups valid                          */
    int64_t n_terms;     /* JJ^T / Jr additions actually made (ok + stale re-adds)  */
} orc_accum_stats;

typedef struct orc_track_stats {
    int32_t iterations;      /* GN iterations executed                       */
    int32_t stopped;         /* 1 if the signed stop rule fired              */
    int32_t nonfinite;       /* 1 if the pose became NaN/inf (reference does not guard) */
    int64_t n_terms_last;    /* orc_accum_stats.n_terms of the last iteration */
    double last_twist[6];
} orc_track_stats;

/* SDF::SDF, sdf.cpp:8-51.  with_global_coords=1 also builds the 24 B/voxel table
 * (sdf.cpp:11,41) that SDF::update reads, as the reference does. */
orc_sdf *orc_sdf_create(int32_t m, float width, float height, float depth,
                        const double origin[3], float delta, float epsilon,
                        int32_t with_global_coords);
void orc_sdf_destroy(orc_sdf *s);
/* start recording the exp() band into s->exp_mask (m^3 bytes, zeroed); 0 on success */
int32_t orc_sdf_track_exp_band(orc_sdf *s);

/* sdf.h:113-127, 132-136, 143-147, 153-157 */
int64_t orc_get_array_index(const orc_sdf *s, const int32_t vox[3]);
void orc_get_voxel_coordinates_idx(const orc_sdf *s, int64_t idx, int32_t vox[3]);
void orc_get_voxel_coordinates(const orc_sdf *s, const double global[3], double vox[3]);
void orc_get_global_coordinates(const orc_sdf *s, const int32_t vox[3], double global[3]);

/* SDF::interpolate_distance, sdf.cpp:127-163 */
float orc_interpolate_distance(const orc_sdf *s, const double vox[3], int32_t *is_interpolated);

/* SDF::create_circle sdf.cpp:100-126 (analytic sphere, W=1) -- used for KATs */
void orc_create_circle(orc_sdf *s, float radius, float cx, float cy, float cz);

/* CameraTracking::CameraTracking, camera_tracking.cpp:3-18 (definition order:
 * max_iter, max_twist_diff, v_h, w_h) */
orc_tracker *orc_tracker_create(int32_t gn_max_iter, float max_twist_diff,
                                float v_h, float w_h, const orc_sdf *s);
void orc_tracker_destroy(orc_tracker *t);
/* camera_tracking.cpp:22-36 */
void orc_tracker_set_K(orc_tracker *t, const double K[9]);
/* camera_tracking.cpp:59-65 */
void orc_set_camera_transformation(orc_tracker *t, const double rot[9], const double trans[3]);

/* cloud helpers: planes are xyz[h*w*3], nrm[h*w*3] (may be NULL), rgb[h*w*3] (may be NULL) */
orc_cloud *orc_cloud_create(int32_t width, int32_t height, const float *xyz,
                            const float *nrm, const uint8_t *rgb);
void orc_cloud_destroy(orc_cloud *c);

/* SDF::update, sdf.cpp:224-315.  with_color=0 skips lines 294-304.
 * threads<=0 -> omp default.  Returns number of voxels updated, or -1 if K missing
 * (reference: exit(0), sdf.cpp:227-230). */
int64_t orc_update(orc_sdf *s, const orc_tracker *t, const orc_cloud *c,
                   int32_t with_color, int32_t threads);

/* CameraTracking::get_partial_derivative, camera_tracking.cpp:246-363.
 * rpm = the six perturbed rotations r1p,r1m,r2p,r2m,r3p,r3m (54 doubles).
 * Returns 0 = out of grid (outputs untouched), 1 = in grid (is_interpolated written). */
int32_t orc_get_partial_derivative(const orc_tracker *t, const orc_sdf *s,
                                   const double rpm[54], const double camera_point[3],
                                   double J[6], int32_t *is_interpolated, double *sdf_val);
/* camera_tracking.cpp:92-145 */
void orc_perturbed_rotations(const orc_tracker *t, double rpm[54]);

/* One Gauss-Newton accumulation = camera_tracking.cpp:81-189 at the tracker's current pose.
 * threads: OpenMP team size (1 = canonical order).  stale_carry=1 reproduces the
 * thread-local carry-over of camera_tracking.cpp:156-159/176-182/261-268; 0 resets
 * the flag per pixel.  Only samples whose centre voxel x-coordinate v satisfies
 * own_x0 <= v < own_x1 contribute (stale re-adds follow their source sample);
 * pass 0, m for everything. */
void orc_accumulate(const orc_tracker *t, const orc_sdf *s, const orc_cloud *c,
                    int32_t threads, int32_t stale_carry, double own_x0, double own_x1,
                    double A[36], double b[6], orc_accum_stats *st);

/* camera_tracking.cpp:191-239 given A,b: solve, exp-map, stop rule, pose update.
 * Returns 1 if the stop rule fired. */
int32_t orc_gn_update(orc_tracker *t, const double A[36], const double b[6], double twist[6]);

/* CameraTracking::estimate_new_position, camera_tracking.cpp:66-245 */
void orc_estimate_new_position(orc_tracker *t, const orc_sdf *s, const orc_cloud *c,
                               int32_t threads, int32_t stale_carry, orc_track_stats *st);

/* ---- mesh extraction (the visualiser thread's work, sdf.cpp:317-391) ------------------------------- */

/* SDF::interpolate_color, sdf.cpp:164-217: inverse-L1 weights over the 8 corners gated by Color_W > 0;
 * the sum is divided by 255 * sum of weights, but an exact hit returns the stored R,G,B undivided, and no
 * valid corner gives 0/0 = NaN.  rgba[3] = 1. */
void orc_interpolate_color(const orc_sdf *s, const double global[3], float rgba[4]);

/* pcl::MarchingCubesSDF::performReconstruction, marching_cubes_sdf.cpp:243-287 (iso level 0 in the
 * reference, sdf.cpp:44): interior voxels (1 <= i,j,k <= m-2, sdf.cpp:36-39) in index order, cube
 * corners/gate of getNeighborList1D (:203-240), createSurface (:100-199) with float arithmetic.
 * Triangulation table: oracle/mc_tables.h (the reference's table, marching_cubes_sdf.h:73-364, kept as data).
 * verts (may be NULL to count): 9 floats per triangle in the grid-local frame of the reference's cloud.
 * Cubes with i in [i0, i1) only (0, m for all).  Returns the number of triangles (those beyond
 * cap_triangles are counted, not written); -1 for an iso level outside [0,1) (:246-252). */
int64_t orc_mesh(const orc_sdf *s, float iso_level, int32_t i0, int32_t i1, float *verts, int64_t cap_triangles);

/* SDF::visualize, sdf.cpp:353-383: world position = cloud point + sdf_origin (double), colour =
 * interpolate_color there.  rgba: 4 floats per vertex. */
void orc_mesh_colors(const orc_sdf *s, const float *verts, int64_t n_vertices, float *rgba);

/* eigen_utils::direct_exponential_map, eigen_utils.cpp:85-128.  out = 3x4 row-major [R|t] */
void orc_direct_exponential_map(const double v[6], double delta_t, double out[12]);
/* Eigen restatements exposed for tests */
void orc_inverse3(const double m[9], double out[9]);
int32_t orc_inverse6(const double A[36], double out[36]);

#ifdef __cplusplus
}
#endif
#endif /* TSDF_ORACLE_H_ */
