/*
 * tsdf_oracle.c -- CPU restatement ("oracle") of tracking_sdf's per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see tsdf_oracle.h).  PARITY UNPINNED vs the real
 * reference binary (no reference golden vectors exist; reference unbuildable
 * here); pinned by hand-derived KATs and a second restatement in NumPy
 * (oracle/np_oracle.py, cross-checked in tests/test_np_oracle.py).
 *
 * Build:  gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC
 * (x86-64 SSE2: float expressions are evaluated in float, no FMA contraction,
 * which is what the reference's flag-less g++ build does, CMakeLists.txt:97-98).
 *
 * Every function cites the reference lines it follows, relative to
 * /root/reference/src/.
 */
#include "tsdf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* (int) of a float / double as x86-64 cvttss2si / cvttsd2si do it: values that
 * do not fit (and NaN) give INT_MIN, the "integer indefinite".  The reference
 * relies on this implicitly (sdf.cpp:143-145, 251-252). */
static inline int32_t trunc_f32(float f) {
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}
static inline int32_t trunc_f64(double f) {
    if (!(f > -2147483649.0 && f < 2147483648.0)) return INT32_MIN;
    return (int32_t)f;
}

/* Fixed-size 3x3 * 3 and 3x3 * 3x3 products (camera_tracking.cpp:51-58, :92-145, :237-238, :40-47; sdf.cpp:245).
 * Eigen 3.2 (the default here: the reference's build era) evaluates a coefficient of a small fixed-size product
 * sequentially, ((a0*b0 + a1*b1) + a2*b2).  Eigen >= 3.3 sends the same expressions through the lazy-product
 * evaluator, whose coefficient is (lhs.row(i).transpose().cwiseProduct(rhs.col(j))).sum(), i.e. the redux unroller's
 * a0*b0 + (a1*b1 + a2*b2).  A reference rebuilt with a modern Eigen therefore differs in last bits of every camera
 * coordinate; orc_set_eigen_order(33) restates that build so that the difference can be quantified
 * (tests/test_eigen_order.py).  The HIP path implements the 3.2 order only.
 * Process-wide switch: test infrastructure, not thread-safe against concurrent oracle calls. */
static int g_eigen_redux_products = 0;
void orc_set_eigen_order(int32_t version) { g_eigen_redux_products = version >= 33; }
int32_t orc_get_eigen_order(void) { return g_eigen_redux_products ? 33 : 32; }
static inline double prod3(double a0, double b0, double a1, double b1, double a2, double b2) {
    return g_eigen_redux_products ? a0 * b0 + (a1 * b1 + a2 * b2) : (a0 * b0 + a1 * b1) + a2 * b2;
}
static inline void mat3_vec(const double M[9], const double v[3], double out[3]) {
    for (int r = 0; r < 3; ++r)
        out[r] = prod3(M[3 * r + 0], v[0], M[3 * r + 1], v[1], M[3 * r + 2], v[2]);
}
static inline void mat3_mat3(const double A[9], const double B[9], double out[9]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            out[3 * r + c] = prod3(A[3 * r + 0], B[0 + c], A[3 * r + 1], B[3 + c], A[3 * r + 2], B[6 + c]);
}
/* Eigen redux unroller for a 3-term sum: x0 + (x1 + x2). */
static inline double dot3_redux(const double a[3], const double b[3]) {
    return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]);
}

/* Matrix3d::inverse(), Eigen cofactor formula (used at camera_tracking.cpp:62). */
static inline double cof3(const double m[9], int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}
void orc_inverse3(const double m[9], double out[9]) {
    double c0[3] = {cof3(m, 0, 0), cof3(m, 1, 0), cof3(m, 2, 0)};
    double col0[3] = {m[0], m[3], m[6]};
    double det = dot3_redux(c0, col0);
    double invdet = 1.0 / det;
    out[0] = c0[0] * invdet; out[1] = c0[1] * invdet; out[2] = c0[2] * invdet;
    out[3] = cof3(m, 0, 1) * invdet; out[4] = cof3(m, 1, 1) * invdet; out[5] = cof3(m, 2, 1) * invdet;
    out[6] = cof3(m, 0, 2) * invdet; out[7] = cof3(m, 1, 2) * invdet; out[8] = cof3(m, 2, 2) * invdet;
}

/* Matrix<double,6,6>::inverse() (camera_tracking.cpp:191): Eigen uses PartialPivLU
 * for sizes > 4: unblocked right-looking LU with row pivoting (first maximum wins),
 * then inverse = solve(Identity).  Returns 0 if a zero pivot was met (result then
 * holds inf/NaN, as Eigen's would). */
int32_t orc_inverse6(const double A[36], double out[36]) {
    double lu[36];
    int perm[6];
    int32_t ok = 1;
    memcpy(lu, A, sizeof lu);
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(lu[6 * k + k]);
        for (int r = k + 1; r < 6; ++r) {
            double a = fabs(lu[6 * r + k]);
            if (a > best) { best = a; piv = r; }
        }
        perm[k] = piv;
        if (best != 0.0) {
            if (piv != k)
                for (int c = 0; c < 6; ++c) {
                    double t = lu[6 * k + c]; lu[6 * k + c] = lu[6 * piv + c]; lu[6 * piv + c] = t;
                }
            for (int r = k + 1; r < 6; ++r) lu[6 * r + k] /= lu[6 * k + k];
        } else {
            ok = 0;
        }
        for (int r = k + 1; r < 6; ++r)
            for (int c = k + 1; c < 6; ++c) lu[6 * r + c] -= lu[6 * r + k] * lu[6 * k + c];
    }
    /* dst = P * I */
    for (int i = 0; i < 36; ++i) out[i] = 0.0;
    for (int i = 0; i < 6; ++i) out[6 * i + i] = 1.0;
    for (int k = 0; k < 6; ++k)
        if (perm[k] != k)
            for (int c = 0; c < 6; ++c) {
                double t = out[6 * k + c]; out[6 * k + c] = out[6 * perm[k] + c]; out[6 * perm[k] + c] = t;
            }
    /* unit-lower forward substitution, then upper back substitution, per column */
    for (int c = 0; c < 6; ++c) {
        for (int r = 0; r < 6; ++r)
            for (int k = 0; k < r; ++k) out[6 * r + c] -= lu[6 * r + k] * out[6 * k + c];
        for (int r = 5; r >= 0; --r) {
            for (int k = r + 1; k < 6; ++k) out[6 * r + c] -= lu[6 * r + k] * out[6 * k + c];
            out[6 * r + c] /= lu[6 * r + r];
        }
    }
    return ok;
}

/* ------------------------------------------------------------------ grid core */

/* sdf.h:113-127 */
int64_t orc_get_array_index(const orc_sdf *s, const int32_t v[3]) {
    if (v[0] < 0 || v[1] < 0 || v[2] < 0) return -1;
    if (v[0] >= s->m || v[1] >= s->m || v[2] >= s->m) return -1;
    int64_t idx = (int64_t)s->m_squared * v[0] + (int64_t)s->m * v[1] + v[2];
    if (idx < 0 || idx >= s->number_of_voxels) return -1;
    return idx;
}
/* sdf.h:132-136 */
void orc_get_voxel_coordinates_idx(const orc_sdf *s, int64_t idx, int32_t v[3]) {
    v[1] = (int32_t)((idx % s->m_squared) / s->m);
    v[0] = (int32_t)(idx / s->m_squared);
    v[2] = (int32_t)(idx % s->m);
}
/* sdf.h:143-147: f64 coordinate times the *float* m_div_*, minus 0.5 */
void orc_get_voxel_coordinates(const orc_sdf *s, const double g[3], double v[3]) {
    v[0] = ((g[0] - s->sdf_origin[0]) * s->m_div_width - 0.5);
    v[1] = ((g[1] - s->sdf_origin[1]) * s->m_div_height - 0.5);
    v[2] = ((g[2] - s->sdf_origin[2]) * s->m_div_depth - 0.5);
}
/* sdf.h:153-157: (extent/(float)m) is a float quotient, widened, times (int+0.5) */
void orc_get_global_coordinates(const orc_sdf *s, const int32_t v[3], double g[3]) {
    g[0] = (s->width / ((float)s->m)) * (v[0] + 0.5) + s->sdf_origin[0];
    g[1] = (s->height / ((float)s->m)) * (v[1] + 0.5) + s->sdf_origin[1];
    g[2] = (s->depth / ((float)s->m)) * (v[2] + 0.5) + s->sdf_origin[2];
}

/* sdf.cpp:8-42 */
orc_sdf *orc_sdf_create(int32_t m, float width, float height, float depth,
                        const double origin[3], float delta, float epsilon,
                        int32_t with_global_coords) {
    orc_sdf *s = (orc_sdf *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->m = m; s->width = width; s->height = height; s->depth = depth;
    s->distance_delta = delta; s->distance_epsilon = epsilon;
    memcpy(s->sdf_origin, origin, sizeof s->sdf_origin);
    s->number_of_voxels = (int64_t)m * m * m;
    s->m_squared = m * m;
    s->m_div_height = m / height;   /* sdf.cpp:19-21: int / float -> float */
    s->m_div_width = m / width;
    s->m_div_depth = m / depth;
    size_t n = (size_t)s->number_of_voxels;
    s->D = (float *)malloc(n * sizeof(float));
    s->W = (float *)malloc(n * sizeof(float));
    s->Color_W = (float *)malloc(n * sizeof(float));
    s->R = (float *)malloc(n * sizeof(float));
    s->G = (float *)malloc(n * sizeof(float));
    s->B = (float *)malloc(n * sizeof(float));
    s->global_coords = with_global_coords ? (double *)malloc(n * 3 * sizeof(double)) : NULL;
    if (!s->D || !s->W || !s->Color_W || !s->R || !s->G || !s->B ||
        (with_global_coords && !s->global_coords)) {
        orc_sdf_destroy(s);
        return NULL;
    }
    const float d0 = width + height + depth;  /* sdf.cpp:29 */
#pragma omp parallel for
    for (int64_t i = 0; i < s->number_of_voxels; ++i) {
        s->D[i] = d0;
        s->Color_W[i] = 0;
        s->W[i] = 0;
        s->R[i] = 0.4;   /* double literal narrowed to float, sdf.cpp:32-34 */
        s->G[i] = 0.4;
        s->B[i] = 0.4;
        if (s->global_coords) {
            int32_t v[3];
            orc_get_voxel_coordinates_idx(s, i, v);
            orc_get_global_coordinates(s, v, &s->global_coords[3 * i]);
        }
    }
    return s;
}
void orc_sdf_destroy(orc_sdf *s) {
    if (!s) return;
    free(s->D); free(s->W); free(s->Color_W); free(s->R); free(s->G); free(s->B);
    free(s->global_coords);
    free(s->exp_mask);
    free(s);
}
int32_t orc_sdf_track_exp_band(orc_sdf *s) {
    if (!s) return -1;
    if (!s->exp_mask) s->exp_mask = (uint8_t *)calloc((size_t)s->number_of_voxels, 1);
    return s->exp_mask ? 0 : -1;
}

/* sdf.cpp:100-126 */
void orc_create_circle(orc_sdf *s, float radius, float cx, float cy, float cz) {
    for (int64_t idx = 0; idx < s->number_of_voxels; ++idx) {
        int32_t v[3]; double g[3];
        orc_get_voxel_coordinates_idx(s, idx, v);
        orc_get_global_coordinates(s, v, g);
        double x = g[0], y = g[1], z = g[2];
        double d = sqrt((x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz));
        s->D[idx] = d - radius;
        s->W[idx] = 1.0;
        s->R[idx] = 0.0;
        s->G[idx] = 0.0;
        s->B[idx] = (x / s->width);
        if (s->B[idx] > 1) s->B[idx] = 1.0;
        if (s->B[idx] < 0.0) s->B[idx] = 0.0;
    }
}

/* sdf.cpp:127-163.  Inverse-L1 weights over the 8 corners, W>0 mask, exact-hit
 * early return, all accumulation in float. */
float orc_interpolate_distance(const orc_sdf *s, const double vox[3], int32_t *is_interpolated) {
    float i = vox[0];   /* f64 -> f32 narrowing, sdf.cpp:130-132 */
    float j = vox[1];
    float k = vox[2];
    float w_sum = 0.0;
    float sum_d = 0.0;
    int32_t cv[3];
    float w = 0;
    float volume;
    int64_t a_idx;
    *is_interpolated = 0;
    for (int io = 0; io < 2; io++) {
        for (int jo = 0; jo < 2; jo++) {
            for (int ko = 0; ko < 2; ko++) {
                /* INT_MIN + 1 stays negative; INT_MIN itself only arises for huge/NaN input */
                cv[0] = trunc_f32(i) + io;
                cv[1] = trunc_f32(j) + jo;
                cv[2] = trunc_f32(k) + ko;
                /* sdf.cpp:146 writes fabs(int - float).  That is FLOAT arithmetic, not C's double fabs: sdf.cpp sees
                 * sdf.h:29 -> camera_tracking.h:11 `using namespace std;` before this line, so the call resolves to
                 * the std::fabs(float) overload (the int operand is converted to float first, the difference, the
                 * absolute value and both additions are float).  Do not "fix" this to fabs(): it changes bits. */
                volume = fabsf(cv[0] - i) + fabsf(cv[1] - j) + fabsf(cv[2] - k);
                a_idx = orc_get_array_index(s, cv);
                if (a_idx != -1) {
                    if (s->W[a_idx] > 0) {
                        *is_interpolated = 1;
                        if (volume < 0.00001) {   /* float widened, compared to a double literal */
                            return s->D[a_idx];
                        }
                        w = 1.0 / volume;         /* double quotient narrowed == float quotient */
                        w_sum += w;
                        sum_d += w * s->D[a_idx];
                    }
                }
            }
        }
    }
    return sum_d / w_sum;
}

/* ------------------------------------------------------------------ tracker state */

/* camera_tracking.cpp:59-65 */
void orc_set_camera_transformation(orc_tracker *t, const double rot[9], const double trans[3]) {
    double r[9], tr[3], tmp[3];
    memcpy(r, rot, sizeof r); memcpy(tr, trans, sizeof tr);   /* inputs may alias t-> */
    memcpy(t->rot, r, sizeof r);
    orc_inverse3(r, t->rot_inv);
    memcpy(t->trans, tr, sizeof tr);
    mat3_vec(t->rot_inv, tr, tmp);
    for (int a = 0; a < 3; ++a) t->rot_inv_trans[a] = -1 * tmp[a];
}

/* camera_tracking.cpp:3-18 */
orc_tracker *orc_tracker_create(int32_t gn_max_iter, float max_twist_diff,
                                float v_h, float w_h, const orc_sdf *s) {
    orc_tracker *t = (orc_tracker *)calloc(1, sizeof *t);
    if (!t) return NULL;
    const double trans0[3] = {0, 0, 1};
    const double rot0[9] = {1, 0, 0, 0, 0, -1, 0, -1, 0};
    orc_set_camera_transformation(t, rot0, trans0);
    t->maximum_twist_diff = max_twist_diff;
    t->gauss_newton_max_iteration = gn_max_iter;
    t->v_h = v_h;
    t->w_h = w_h;
    t->v_h2 = 2 * v_h;
    t->w_h2 = 2 * w_h;
    t->v_h2_width = t->v_h2 / s->m_div_width;
    t->v_h2_height = t->v_h2 / s->m_div_height;
    t->v_h2_depth = t->v_h2 / s->m_div_depth;
    t->isKFilled = 0;
    return t;
}
void orc_tracker_destroy(orc_tracker *t) { free(t); }

/* camera_tracking.cpp:22-36 */
void orc_tracker_set_K(orc_tracker *t, const double K[9]) {
    memcpy(t->K, K, sizeof t->K);
    t->isKFilled = 1;
}

/* ------------------------------------------------------------------ clouds */

orc_cloud *orc_cloud_create(int32_t width, int32_t height, const float *xyz,
                            const float *nrm, const uint8_t *rgb) {
    orc_cloud *c = (orc_cloud *)calloc(1, sizeof *c);
    if (!c) return NULL;
    size_t n = (size_t)width * height;
    c->width = width; c->height = height;
    c->points = (orc_point *)calloc(n, sizeof(orc_point));
    c->normals = (orc_normal *)calloc(n, sizeof(orc_normal));
    if (!c->points || !c->normals) { orc_cloud_destroy(c); return NULL; }
    for (size_t p = 0; p < n; ++p) {
        c->points[p].x = xyz[3 * p + 0];
        c->points[p].y = xyz[3 * p + 1];
        c->points[p].z = xyz[3 * p + 2];
        if (rgb) { c->points[p].r = rgb[3 * p + 0]; c->points[p].g = rgb[3 * p + 1]; c->points[p].b = rgb[3 * p + 2]; }
        if (nrm) { c->normals[p].nx = nrm[3 * p + 0]; c->normals[p].ny = nrm[3 * p + 1]; c->normals[p].nz = nrm[3 * p + 2]; }
        else { c->normals[p].nx = c->normals[p].ny = c->normals[p].nz = NAN; }
    }
    return c;
}
void orc_cloud_destroy(orc_cloud *c) {
    if (!c) return;
    free(c->points); free(c->normals); free(c);
}

/* ------------------------------------------------------------------ SDF::update */

/* sdf.cpp:224-315 */
int64_t orc_update(orc_sdf *s, const orc_tracker *t, const orc_cloud *c,
                   int32_t with_color, int32_t threads) {
    if (!t->isKFilled) return -1;   /* sdf.cpp:227-230 (reference: exit(0)) */
    int64_t n_updated = 0;
#ifdef _OPENMP
    int np = threads > 0 ? threads : omp_get_max_threads();
#else
    int np = 1; (void)threads;
#endif
#pragma omp parallel for num_threads(np) reduction(+ : n_updated)
    for (int64_t idx = 0; idx < s->number_of_voxels; idx++) {
        double global_coordinates[3], camera_point[3], camera_point_img[3], normal_eigen[3], ij[3], tmp[3];
        double image_point[2];
        float d_new;
        float w_new, w_old;
        int32_t i_image, j_image;

        if (s->global_coords) {                                  /* sdf.cpp:244 */
            global_coordinates[0] = s->global_coords[3 * idx + 0];
            global_coordinates[1] = s->global_coords[3 * idx + 1];
            global_coordinates[2] = s->global_coords[3 * idx + 2];
        } else {
            int32_t v[3];
            orc_get_voxel_coordinates_idx(s, idx, v);
            orc_get_global_coordinates(s, v, global_coordinates);
        }
        /* project_world_to_camera, camera_tracking.cpp:51-54 */
        mat3_vec(t->rot_inv, global_coordinates, tmp);
        camera_point[0] = tmp[0] + t->rot_inv_trans[0];
        camera_point[1] = tmp[1] + t->rot_inv_trans[1];
        camera_point[2] = tmp[2] + t->rot_inv_trans[2];
        if (camera_point[2] < 0) continue;                       /* sdf.cpp:247-249 */
        /* project_camera_to_image_plane, camera_tracking.cpp:40-47 */
        mat3_vec(t->K, camera_point, ij);
        image_point[0] = ij[0] / ij[2];
        image_point[1] = ij[1] / ij[2];
        i_image = trunc_f64(image_point[0]);                     /* sdf.cpp:251-252 */
        j_image = trunc_f64(image_point[1]);
        /* sdf.cpp:254: int compared with uint32 width => negatives are rejected */
        if ((uint32_t)i_image >= (uint32_t)c->width || (uint32_t)j_image >= (uint32_t)c->height ||
            i_image < 0 || j_image < 0)
            continue;
        const orc_point point = c->points[(size_t)j_image * c->width + i_image];    /* at(col,row) */
        const orc_normal normal = c->normals[(size_t)j_image * c->width + i_image];
        if (isnan(point.x) || isnan(point.y) || isnan(normal.nx) || isnan(normal.ny) || isnan(normal.nz))
            continue;                                            /* sdf.cpp:260-262 */
        camera_point_img[0] = point.x;
        camera_point_img[1] = point.y;
        camera_point_img[2] = point.z;
        normal_eigen[0] = normal.nx;
        normal_eigen[1] = normal.ny;
        normal_eigen[2] = normal.nz;
        /* projectivePointToPlaneDistance, sdf.h:177-181 */
        double diff_vec[3] = {camera_point_img[0] - camera_point[0],
                              camera_point_img[1] - camera_point[1],
                              camera_point_img[2] - camera_point[2]};
        double pointToPlaneDistance = dot3_redux(diff_vec, normal_eigen);
        d_new = pointToPlaneDistance;                            /* sdf.cpp:274, f64 -> f32 */
        w_new = 1.0;
        if (d_new >= s->distance_epsilon && d_new <= s->distance_delta) {
            w_new = exp(-0.5 * (d_new - s->distance_epsilon) * (d_new - s->distance_epsilon));
            if (s->exp_mask) s->exp_mask[idx] = 1;               /* (checker's annotation, not the reference's) */
        }
        if (d_new > s->distance_delta) continue;                 /* sdf.cpp:280-283 */
        if (d_new < -s->distance_delta) d_new = -s->distance_delta;

        w_old = s->W[idx];
        s->W[idx] = w_old + w_new;
        s->D[idx] = (w_old * s->D[idx] + w_new * d_new) / s->W[idx];
        n_updated++;

        if (with_color) {                                        /* sdf.cpp:294-304 */
            const double cam_vect[3] = {0, 0, 1};
            double cosine = fabs(dot3_redux(cam_vect, normal_eigen)) /
                            sqrt(dot3_redux(normal_eigen, normal_eigen));
            w_old = s->Color_W[idx];
            w_new = w_new * cosine;
            s->Color_W[idx] = w_old + w_new;
            s->R[idx] = (w_old * s->R[idx] + w_new * point.r) / s->Color_W[idx];
            s->G[idx] = (w_old * s->G[idx] + w_new * point.g) / s->Color_W[idx];
            s->B[idx] = (w_old * s->B[idx] + w_new * point.b) / s->Color_W[idx];
        }
    }
    return n_updated;
}

/* ------------------------------------------------------------------ tracker */

/* camera_tracking.cpp:92-145: r_k+- = (I +- w_h [e_k]x) * rot, w_h float widened */
void orc_perturbed_rotations(const orc_tracker *t, double rpm[54]) {
    const double wh = t->w_h;
    double Rd[9];
    for (int k = 0; k < 3; ++k) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            const double s = sgn == 0 ? wh : -wh;
            for (int e = 0; e < 9; ++e) Rd[e] = 0.0;
            Rd[0] = Rd[4] = Rd[8] = 1.0;
            if (k == 0) { Rd[5] = -s; Rd[7] = s; }        /* (1,2)=-w (2,1)=+w */
            else if (k == 1) { Rd[2] = s; Rd[6] = -s; }   /* (0,2)=+w (2,0)=-w */
            else { Rd[1] = -s; Rd[3] = s; }               /* (0,1)=-w (1,0)=+w */
            mat3_mat3(Rd, t->rot, &rpm[9 * (2 * k + sgn)]);
        }
    }
}

/* camera_tracking.cpp:246-363 */
int32_t orc_get_partial_derivative(const orc_tracker *t, const orc_sdf *s,
                                   const double rpm[54], const double camera_point[3],
                                   double J[6], int32_t *is_interpolated, double *sdf_val) {
    double cw[3], cv[3], pw[3], mw[3], pv[3], mv[3], tmp[3];
    float plus_v, minus_v;
    /* project_camera_to_world, camera_tracking.cpp:55-58 */
    mat3_vec(t->rot, camera_point, tmp);
    cw[0] = tmp[0] + t->trans[0]; cw[1] = tmp[1] + t->trans[1]; cw[2] = tmp[2] + t->trans[2];
    orc_get_voxel_coordinates(s, cw, cv);
    if (cv[0] < 0 || cv[1] < 0 || cv[2] < 0) return 0;                 /* :261-264 */
    if (cv[0] >= s->m || cv[1] >= s->m || cv[2] >= s->m) return 0;     /* :265-268 */
    *sdf_val = orc_interpolate_distance(s, cv, is_interpolated);        /* :269 */
    if (!*is_interpolated) return 1;

    const float vh2[3] = {t->v_h2_width, t->v_h2_height, t->v_h2_depth};
    for (int a = 0; a < 3; ++a) {                                       /* :273-316 */
        memcpy(pv, cv, sizeof pv); memcpy(mv, cv, sizeof mv);
        pv[a] += t->v_h;
        mv[a] -= t->v_h;
        plus_v = orc_interpolate_distance(s, pv, is_interpolated);
        if (!*is_interpolated) return 1;
        minus_v = orc_interpolate_distance(s, mv, is_interpolated);
        if (!*is_interpolated) return 1;
        J[a] = (plus_v - minus_v) / vh2[a];          /* float quotient widened */
    }
    for (int a = 0; a < 3; ++a) {                                       /* :318-361 */
        mat3_vec(&rpm[9 * (2 * a + 0)], camera_point, tmp);
        pw[0] = tmp[0] + t->trans[0]; pw[1] = tmp[1] + t->trans[1]; pw[2] = tmp[2] + t->trans[2];
        mat3_vec(&rpm[9 * (2 * a + 1)], camera_point, tmp);
        mw[0] = tmp[0] + t->trans[0]; mw[1] = tmp[1] + t->trans[1]; mw[2] = tmp[2] + t->trans[2];
        orc_get_voxel_coordinates(s, pw, pv);
        orc_get_voxel_coordinates(s, mw, mv);
        plus_v = orc_interpolate_distance(s, pv, is_interpolated);
        if (!*is_interpolated) return 1;
        minus_v = orc_interpolate_distance(s, mv, is_interpolated);
        if (!*is_interpolated) return 1;
        J[3 + a] = (plus_v - minus_v) / (2 * (t->w_h));   /* float quotient widened */
    }
    return 1;
}

/* camera_tracking.cpp:81-189 */
void orc_accumulate(const orc_tracker *t, const orc_sdf *s, const orc_cloud *c,
                    int32_t threads, int32_t stale_carry, double own_x0, double own_x1,
                    double A[36], double b[6], orc_accum_stats *st) {
    double rpm[54];
    orc_perturbed_rotations(t, rpm);
#ifdef _OPENMP
    int np = threads > 0 ? threads : omp_get_max_threads();
#else
    int np = 1; (void)threads;
#endif
    double *A_array = (double *)calloc((size_t)np * 36, sizeof(double));
    double *B_array = (double *)calloc((size_t)np * 6, sizeof(double));
    orc_accum_stats *S_array = (orc_accum_stats *)calloc((size_t)np, sizeof(orc_accum_stats));
    const int ncols = (c->width + 2) / 3;
#pragma omp parallel num_threads(np)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        double *A_t = &A_array[36 * tid], *B_t = &B_array[6 * tid];
        orc_accum_stats *S = &S_array[tid];
        /* thread-locals that live across pixels: camera_tracking.cpp:156-159 */
        int32_t is_interpolated = 0;
        double J[6] = {0, 0, 0, 0, 0, 0};
        double int_dist = 0;
        int32_t owned = 0;   /* ownership of the sample that last wrote J / int_dist */
#pragma omp for
        for (int ci = 0; ci < ncols; ++ci) {                   /* :162  i += 3 */
            const int i = 3 * ci;
            for (int j = 0; j < c->height; j += 3) {           /* :163 */
                const orc_point point = c->points[(size_t)j * c->width + i];
                S->n_samples++;
                if (isnan(point.x) || isnan(point.y) || isnan(point.z)) { S->n_nan++; continue; }
                const double cp[3] = {point.x, point.y, point.z};
                if (!stale_carry) is_interpolated = 0;
                /* ownership uses the same centre voxel coordinate the reference computes */
                double tmp[3], cw[3], cv[3];
                mat3_vec(t->rot, cp, tmp);
                cw[0] = tmp[0] + t->trans[0]; cw[1] = tmp[1] + t->trans[1]; cw[2] = tmp[2] + t->trans[2];
                orc_get_voxel_coordinates(s, cw, cv);
                const int32_t in_grid =
                    orc_get_partial_derivative(t, s, rpm, cp, J, &is_interpolated, &int_dist);
                if (in_grid) {
                    owned = (cv[0] >= own_x0 && cv[0] < own_x1);
                    if (is_interpolated) S->n_ok++; else S->n_fail++;
                } else {
                    S->n_oog++;
                }
                if (!is_interpolated) continue;                 /* :178 */
                if (!owned) continue;
                for (int r = 0; r < 6; ++r)                     /* :181 */
                    for (int q = 0; q < 6; ++q) A_t[6 * r + q] = A_t[6 * r + q] + (J[r] * J[q]);
                for (int r = 0; r < 6; ++r) B_t[r] = B_t[r] + (int_dist * J[r]);   /* :182 */
                S->n_terms++;
            }
        }
    }
    for (int e = 0; e < 36; ++e) A[e] = 0.0;
    for (int e = 0; e < 6; ++e) b[e] = 0.0;
    orc_accum_stats tot; memset(&tot, 0, sizeof tot);
    for (int p = 0; p < np; ++p) {                              /* :186-189 */
        for (int e = 0; e < 36; ++e) A[e] += A_array[36 * p + e];
        for (int e = 0; e < 6; ++e) b[e] += B_array[6 * p + e];
        tot.n_samples += S_array[p].n_samples; tot.n_nan += S_array[p].n_nan;
        tot.n_oog += S_array[p].n_oog; tot.n_fail += S_array[p].n_fail;
        tot.n_ok += S_array[p].n_ok; tot.n_terms += S_array[p].n_terms;
    }
    if (st) *st = tot;
    free(A_array); free(B_array); free(S_array);
}

/* eigen_utils.cpp:40-59 */
static const double ang_min_sinc = 1.0e-8;
static const double ang_min_mc = 2.5e-4;
static double f_sinc(double sinx, double x) { return fabs(x) < ang_min_sinc ? 1.0 : (sinx / x); }
static double f_mcosc(double cosx, double x) { return fabs(x) < ang_min_mc ? 0.5 : ((1.0 - cosx) / x / x); }
static double f_msinc(double sinx, double x) { return fabs(x) < ang_min_mc ? (1. / 6.0) : ((1.0 - sinx / x) / x / x); }

/* eigen_utils.cpp:61-128 */
void orc_direct_exponential_map(const double v[6], double delta_t, double out[12]) {
    double v_dt[6], u[3];
    for (int a = 0; a < 6; ++a) v_dt[a] = v[a] * delta_t;
    u[0] = v_dt[3]; u[1] = v_dt[4]; u[2] = v_dt[5];
    double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double si = sin(theta), co = cos(theta);
    double sinc = f_sinc(si, theta), mcosc = f_mcosc(co, theta), msinc = f_msinc(si, theta);
    double rd[9];
    rd[0] = co + mcosc * u[0] * u[0];
    rd[1] = -sinc * u[2] + mcosc * u[0] * u[1];
    rd[2] = sinc * u[1] + mcosc * u[0] * u[2];
    rd[3] = sinc * u[2] + mcosc * u[1] * u[0];
    rd[4] = co + mcosc * u[1] * u[1];
    rd[5] = -sinc * u[0] + mcosc * u[1] * u[2];
    rd[6] = -sinc * u[1] + mcosc * u[2] * u[0];
    rd[7] = sinc * u[0] + mcosc * u[2] * u[1];
    rd[8] = co + mcosc * u[2] * u[2];
    double dt[3];
    dt[0] = v_dt[0] * (sinc + u[0] * u[0] * msinc)
          + v_dt[1] * (u[0] * u[1] * msinc - u[2] * mcosc)
          + v_dt[2] * (u[0] * u[2] * msinc + u[1] * mcosc);
    dt[1] = v_dt[0] * (u[0] * u[1] * msinc + u[2] * mcosc)
          + v_dt[1] * (sinc + u[1] * u[1] * msinc)
          + v_dt[2] * (u[1] * u[2] * msinc - u[0] * mcosc);
    dt[2] = v_dt[0] * (u[0] * u[2] * msinc - u[1] * mcosc)
          + v_dt[1] * (u[1] * u[2] * msinc + u[0] * mcosc)
          + v_dt[2] * (sinc + u[2] * u[2] * msinc);
    for (int r = 0; r < 3; ++r) {
        out[4 * r + 0] = rd[3 * r + 0]; out[4 * r + 1] = rd[3 * r + 1]; out[4 * r + 2] = rd[3 * r + 2];
        out[4 * r + 3] = dt[r];
    }
}

/* camera_tracking.cpp:191-239 */
int32_t orc_gn_update(orc_tracker *t, const double A[36], const double b[6], double twist[6]) {
    double Ainv[36];
    orc_inverse6(A, Ainv);
    for (int r = 0; r < 6; ++r) {                               /* :191, sequential 6-term rows */
        double acc = Ainv[6 * r + 0] * b[0];
        for (int k = 1; k < 6; ++k) acc += Ainv[6 * r + k] * b[k];
        twist[r] = acc;
    }
    double aff[12];
    orc_direct_exponential_map(twist, 1.0, aff);                /* :192 */
    const double mtd = t->maximum_twist_diff;                   /* float widened */
    int32_t stop = (twist[0] < mtd && twist[1] < mtd && twist[2] < mtd &&
                    twist[3] < mtd && twist[4] < mtd && twist[5] < mtd);   /* :216-224, signed */
    /* aff.rotation() == linear block for an orthogonal input (polar factor, ~1e-16) */
    double Rt[9], at[3], newrot[9], tmp[3], newtrans[3];
    for (int r = 0; r < 3; ++r) {
        for (int cc = 0; cc < 3; ++cc) Rt[3 * r + cc] = aff[4 * cc + r];
        at[r] = aff[4 * r + 3];
    }
    mat3_mat3(Rt, t->rot, newrot);                              /* :237 */
    mat3_vec(Rt, at, tmp);                                      /* :238 */
    for (int a = 0; a < 3; ++a) newtrans[a] = t->trans[a] - tmp[a];
    orc_set_camera_transformation(t, newrot, newtrans);         /* :239 */
    return stop;
}

/* camera_tracking.cpp:66-245 */
void orc_estimate_new_position(orc_tracker *t, const orc_sdf *s, const orc_cloud *c,
                               int32_t threads, int32_t stale_carry, orc_track_stats *st) {
    int32_t stop = 0, g = 0;
    double A[36], b[6], twist[6] = {0, 0, 0, 0, 0, 0};
    orc_accum_stats as; memset(&as, 0, sizeof as);
    for (g = 0; g < t->gauss_newton_max_iteration && !stop; g++) {
        orc_accumulate(t, s, c, threads, stale_carry, 0.0, (double)s->m, A, b, &as);
        stop = orc_gn_update(t, A, b, twist);
    }
    if (st) {
        st->iterations = g;
        st->stopped = stop;
        st->n_terms_last = as.n_terms;
        st->nonfinite = 0;
        for (int a = 0; a < 9; ++a) if (!isfinite(t->rot[a])) st->nonfinite = 1;
        for (int a = 0; a < 3; ++a) if (!isfinite(t->trans[a])) st->nonfinite = 1;
        memcpy(st->last_twist, twist, sizeof twist);
    }
}

/* ------------------------------------------------------------------ mesh extraction */

#include "mc_tables.h"

/* sdf.cpp:164-217 */
void orc_interpolate_color(const orc_sdf *s, const double global[3], float rgba[4]) {
    double vc[3];
    orc_get_voxel_coordinates(s, global, vc);
    float i = vc[0];
    float j = vc[1];
    float k = vc[2];
    float w_sum = 0.0;
    float aux = 0;
    float r = 0.0, g = 0.0, b = 0.0;    /* std_msgs::ColorRGBA fields are float32 */
    int32_t cv[3];
    float w = 0;
    float volume;
    int64_t a_idx;
    rgba[3] = 1.0;
    for (int io = 0; io < 2; io++) {
        for (int jo = 0; jo < 2; jo++) {
            for (int ko = 0; ko < 2; ko++) {
                cv[0] = trunc_f32(i) + io;
                cv[1] = trunc_f32(j) + jo;
                cv[2] = trunc_f32(k) + ko;
                volume = fabsf(cv[0] - i) + fabsf(cv[1] - j) + fabsf(cv[2] - k);
                a_idx = orc_get_array_index(s, cv);
                if (a_idx != -1) {
                    if (s->Color_W[a_idx] > 0) {
                        if (volume < 0.00001) {          /* :195-200: stored values, not divided by 255 */
                            rgba[0] = s->R[a_idx]; rgba[1] = s->G[a_idx]; rgba[2] = s->B[a_idx];
                            return;
                        }
                        w = 1.0 / volume;
                        w_sum += w;
                        r += w * s->R[a_idx];
                        g += w * s->G[a_idx];
                        b += w * s->B[a_idx];
                    }
                }
            }
        }
    }
    aux = w_sum * 255.0;      /* double product narrowed, :212 */
    rgba[0] = r / aux;
    rgba[1] = g / aux;
    rgba[2] = b / aux;
}

/* marching_cubes_sdf.cpp:87-94 */
static void mc_interpolate_edge(float iso, const float p1[3], const float p2[3], float v1, float v2, float out[3]) {
    float mu = (iso - v1) / (v2 - v1);
    for (int a = 0; a < 3; ++a) out[a] = p1[a] + mu * (p2[a] - p1[a]);
}

int64_t orc_mesh(const orc_sdf *s, float iso, int32_t i0, int32_t i1, float *verts, int64_t cap) {
    if (!(iso >= 0 && iso < 1)) return -1;                       /* :246-252 */
    static const int edge_a[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};    /* :146-169 */
    static const int edge_b[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    const int m = s->m;
    const int64_t zy = (int64_t)m * m;                           /* zy_index_offset, marching_cubes_sdf.h:406 */
    const float min_p[3] = {0, 0, 0};                            /* setBBox, :55-65 */
    const float max_p[3] = {s->width, s->height, s->depth};
    int64_t n_tri = 0;
    for (int i = 1; i < m - 1; ++i) {                            /* voxel_coords order, sdf.cpp:26-40 */
        if (i < i0 || i >= i1) continue;
        for (int j = 1; j < m - 1; ++j)
            for (int k = 1; k < m - 1; ++k) {
                /* getNeighborList1D, :203-240 */
                const int64_t g0 = (int64_t)i * zy + (int64_t)j * m + k;
                const int64_t g[8] = {g0, g0 + zy, g0 + zy + 1, g0 + 1, g0 + m, g0 + m + zy, g0 + m + zy + 1, g0 + m + 1};
                float leaf[8];
                int all = 1;
                for (int c = 0; c < 8; ++c) all = all && (s->W[g[c]] > 0);
                for (int c = 0; c < 8; ++c) leaf[c] = all ? s->D[g[c]] : s->D[g0];
                /* createSurface, :100-199 */
                int cubeindex = 0;
                for (int c = 0; c < 8; ++c) if (leaf[c] < iso) cubeindex |= 1 << c;
                const int nt = kMcNumTri[cubeindex];
                if (nt == 0) continue;                           /* edgeTable[cubeindex] == 0 */
                const int idx3[3] = {i, j, k};
                const int res[3] = {m, m, m};
                float center[3];
                for (int a = 0; a < 3; ++a)
                    center[a] = min_p[a] + (max_p[a] - min_p[a]) * (float)idx3[a] / (float)res[a];
                float p[8][3];
                for (int c = 0; c < 8; ++c) {
                    p[c][0] = center[0]; p[c][1] = center[1]; p[c][2] = center[2];
                    if (c & 0x4) p[c][1] = (float)(center[1] + (max_p[1] - min_p[1]) / (float)res[1]);
                    if (c & 0x2) p[c][2] = (float)(center[2] + (max_p[2] - min_p[2]) / (float)res[2]);
                    if ((c & 0x1) ^ ((c >> 1) & 0x1)) p[c][0] = (float)(center[0] + (max_p[0] - min_p[0]) / (float)res[0]);
                }
                float vl[12][3];
                for (int e = 0; e < 12; ++e) {
                    const int a = edge_a[e], b = edge_b[e];
                    if ((leaf[a] < iso) != (leaf[b] < iso))      /* bit e of edgeTable[cubeindex] */
                        mc_interpolate_edge(iso, p[a], p[b], leaf[a], leaf[b], vl[e]);
                }
                for (int t = 0; t < nt; ++t) {
                    if (verts && n_tri < cap)
                        for (int v = 0; v < 3; ++v) {
                            const int e = kMcTri[cubeindex][3 * t + v];
                            for (int a = 0; a < 3; ++a) verts[9 * n_tri + 3 * v + a] = vl[e][a];
                        }
                    ++n_tri;
                }
            }
    }
    return n_tri;
}

/* sdf.cpp:353-383 */
void orc_mesh_colors(const orc_sdf *s, const float *verts, int64_t n_vertices, float *rgba) {
    for (int64_t v = 0; v < n_vertices; ++v) {
        double g[3];
        for (int a = 0; a < 3; ++a) g[a] = verts[3 * v + a] + s->sdf_origin[a];
        orc_interpolate_color(s, g, &rgba[4 * v]);
    }
}
