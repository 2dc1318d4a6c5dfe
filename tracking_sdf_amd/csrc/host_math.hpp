// host_math.hpp -- the tracker's host-side algebra: pose bookkeeping, the six perturbed
// rotations, the 6x6 normal-equation solve, the SE(3) exponential map, the stop rule and the pose
// composition.  It runs once per Gauss-Newton iteration on 27 doubles, so it stays on the host.
//
// Mirrors (paths relative to the reference's src/):
//   set_pose              CameraTracking::set_camera_transformation   src/camera_tracking.cpp:59-65
//   perturbed_rotations   src/camera_tracking.cpp:92-145
//   solve_normal          twist = A.inverse() * b                     src/camera_tracking.cpp:191
//   exp_se3               eigen_utils::direct_exponential_map         src/eigen_utils.cpp:40-128
//   gn_step               stop rule + pose update                     src/camera_tracking.cpp:216-239
//
// Evaluation orders follow Eigen 3.2's fixed-size kernels (sequential 3-term products, cofactor
// 3x3 inverse, partial-pivot LU inverse for 6x6); compile with -ffp-contract=off.
#pragma once

#include <cmath>
#include <cstring>

namespace tsdf {
namespace hm {

struct Pose {
    double rot[9];            // camera -> world
    double trans[3];
    double rot_inv[9];        // world -> camera
    double rot_inv_trans[3];
};

inline void mul33v(const double* M, const double* v, double* out) {
    for (int r = 0; r < 3; ++r) out[r] = (M[3 * r] * v[0] + M[3 * r + 1] * v[1]) + M[3 * r + 2] * v[2];
}
inline void mul33(const double* A, const double* B, double* out) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            out[3 * r + c] = (A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c]) + A[3 * r + 2] * B[6 + c];
}

// Cofactor inverse of a 3x3 (the reference inverts `rot` instead of transposing it; its initial
// "rotation" has det = -1, camera_tracking.cpp:7, so the two are not interchangeable in general).
inline void inverse33(const double* m, double* out) {
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
    };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = c00 * m[0] + (c10 * m[3] + c20 * m[6]);
    const double inv = 1.0 / det;
    out[0] = c00 * inv; out[1] = c10 * inv; out[2] = c20 * inv;
    out[3] = cof(0, 1) * inv; out[4] = cof(1, 1) * inv; out[5] = cof(2, 1) * inv;
    out[6] = cof(0, 2) * inv; out[7] = cof(1, 2) * inv; out[8] = cof(2, 2) * inv;
}

inline void set_pose(Pose& P, const double* rot, const double* trans) {
    double r[9], t[3], tmp[3];
    std::memcpy(r, rot, sizeof r);
    std::memcpy(t, trans, sizeof t);
    std::memcpy(P.rot, r, sizeof r);
    inverse33(r, P.rot_inv);
    std::memcpy(P.trans, t, sizeof t);
    mul33v(P.rot_inv, t, tmp);
    for (int a = 0; a < 3; ++a) P.rot_inv_trans[a] = -1 * tmp[a];
}

// r_k+- = (I +- w_h [e_k]x) * rot, order r1p r1m r2p r2m r3p r3m.
inline void perturbed_rotations(const Pose& P, float w_h, double* rpm /*54*/) {
    const double wh = (double)w_h;
    for (int k = 0; k < 3; ++k)
        for (int sgn = 0; sgn < 2; ++sgn) {
            const double s = sgn ? -wh : wh;
            double Rd[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (k == 0) { Rd[5] = -s; Rd[7] = s; }
            else if (k == 1) { Rd[2] = s; Rd[6] = -s; }
            else { Rd[1] = -s; Rd[3] = s; }
            mul33(Rd, P.rot, rpm + 9 * (2 * k + sgn));
        }
}

// x = inverse(A) * b through a partial-pivot LU of A (unblocked, first maximum wins) and
// inverse = solve(I).  Returns false when a pivot is exactly zero or the result is not finite.
inline bool solve_normal(const double* A, const double* b, double* x) {
    double lu[36], inv[36];
    int perm[6];
    bool ok = true;
    std::memcpy(lu, A, sizeof lu);
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = std::fabs(lu[6 * k + k]);
        for (int r = k + 1; r < 6; ++r) {
            const double a = std::fabs(lu[6 * r + k]);
            if (a > best) { best = a; piv = r; }
        }
        perm[k] = piv;
        if (best != 0.0) {
            if (piv != k)
                for (int c = 0; c < 6; ++c) { const double t = lu[6 * k + c]; lu[6 * k + c] = lu[6 * piv + c]; lu[6 * piv + c] = t; }
            for (int r = k + 1; r < 6; ++r) lu[6 * r + k] /= lu[6 * k + k];
        } else {
            ok = false;
        }
        for (int r = k + 1; r < 6; ++r)
            for (int c = k + 1; c < 6; ++c) lu[6 * r + c] -= lu[6 * r + k] * lu[6 * k + c];
    }
    for (int e = 0; e < 36; ++e) inv[e] = 0.0;
    for (int d = 0; d < 6; ++d) inv[6 * d + d] = 1.0;
    for (int k = 0; k < 6; ++k)
        if (perm[k] != k)
            for (int c = 0; c < 6; ++c) { const double t = inv[6 * k + c]; inv[6 * k + c] = inv[6 * perm[k] + c]; inv[6 * perm[k] + c] = t; }
    for (int c = 0; c < 6; ++c) {
        for (int r = 0; r < 6; ++r)
            for (int k = 0; k < r; ++k) inv[6 * r + c] -= lu[6 * r + k] * inv[6 * k + c];
        for (int r = 5; r >= 0; --r) {
            for (int k = r + 1; k < 6; ++k) inv[6 * r + c] -= lu[6 * r + k] * inv[6 * k + c];
            inv[6 * r + c] /= lu[6 * r + r];
        }
    }
    for (int r = 0; r < 6; ++r) {
        double acc = inv[6 * r] * b[0];
        for (int k = 1; k < 6; ++k) acc += inv[6 * r + k] * b[k];
        x[r] = acc;
        if (!std::isfinite(acc)) ok = false;
    }
    return ok;
}

// exp of the twist (v, w) * dt as a 3x4 [R|t], with the reference's small-angle guards.
//
// Bit parity with the reference requires its closed form term for term: this function restates
// eigen_utils::direct_exponential_map / UThetaToAffine3d / f_sinc / f_mcosc / f_msinc of the reference's
// src/src/eigen_utils.cpp:40-128, which is distributed under the following terms (BSD 3-clause):
//
//   Copyright (c) 2013, Willow Garage, Inc.  All rights reserved.
//   Author: Mario Prats.  Much of this code has been adapted from the ViSP library (http://www.irisa.fr/lagadic/visp).
//
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the
//   following conditions are met:
//     * Redistributions of source code must retain the above copyright notice, this list of conditions and the
//       following disclaimer.
//     * Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the
//       following disclaimer in the documentation and/or other materials provided with the distribution.
//     * Neither the name of the Willow Garage, Inc. nor the names of its contributors may be used to endorse or promote
//       products derived from this software without specific prior written permission.
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES,
//   INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
//   DISCLAIMED.  IN NO EVENT SHALL THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL,
//   SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR
//   SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY,
//   WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE
//   OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
inline void exp_se3(const double* xi, double dt, double* R /*9*/, double* t /*3*/) {
    double v[3] = {xi[0] * dt, xi[1] * dt, xi[2] * dt};
    double u[3] = {xi[3] * dt, xi[4] * dt, xi[5] * dt};
    const double theta = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double si = std::sin(theta), co = std::cos(theta);
    const double sinc = std::fabs(theta) < 1.0e-8 ? 1.0 : (si / theta);
    const double mcosc = std::fabs(theta) < 2.5e-4 ? 0.5 : ((1.0 - co) / theta / theta);
    const double msinc = std::fabs(theta) < 2.5e-4 ? (1. / 6.0) : ((1.0 - si / theta) / theta / theta);
    R[0] = co + mcosc * u[0] * u[0];
    R[1] = -sinc * u[2] + mcosc * u[0] * u[1];
    R[2] = sinc * u[1] + mcosc * u[0] * u[2];
    R[3] = sinc * u[2] + mcosc * u[1] * u[0];
    R[4] = co + mcosc * u[1] * u[1];
    R[5] = -sinc * u[0] + mcosc * u[1] * u[2];
    R[6] = -sinc * u[1] + mcosc * u[2] * u[0];
    R[7] = sinc * u[0] + mcosc * u[2] * u[1];
    R[8] = co + mcosc * u[2] * u[2];
    t[0] = v[0] * (sinc + u[0] * u[0] * msinc) + v[1] * (u[0] * u[1] * msinc - u[2] * mcosc) + v[2] * (u[0] * u[2] * msinc + u[1] * mcosc);
    t[1] = v[0] * (u[0] * u[1] * msinc + u[2] * mcosc) + v[1] * (sinc + u[1] * u[1] * msinc) + v[2] * (u[1] * u[2] * msinc - u[0] * mcosc);
    t[2] = v[0] * (u[0] * u[2] * msinc - u[1] * mcosc) + v[1] * (u[1] * u[2] * msinc + u[0] * mcosc) + v[2] * (sinc + u[2] * u[2] * msinc);
}

// One Gauss-Newton step on the pose.  Returns false (pose untouched) if the system is singular or
// the new pose is not finite.  *stop = every SIGNED component of the twist below the threshold.
inline bool gn_step(Pose& P, const double* A, const double* b, float max_twist_diff, double* twist, bool* stop) {
    if (!solve_normal(A, b, twist)) return false;
    double R[9], t[3], Rt[9], nrot[9], tmp[3], ntrans[3];
    exp_se3(twist, 1.0, R, t);
    const double thr = (double)max_twist_diff;
    *stop = twist[0] < thr && twist[1] < thr && twist[2] < thr && twist[3] < thr && twist[4] < thr && twist[5] < thr;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * c + r];
    mul33(Rt, P.rot, nrot);                    // rot <- R^T rot
    mul33v(Rt, t, tmp);
    for (int a = 0; a < 3; ++a) ntrans[a] = P.trans[a] - tmp[a];   // trans <- trans - R^T t
    for (int e = 0; e < 9; ++e) if (!std::isfinite(nrot[e])) return false;
    for (int e = 0; e < 3; ++e) if (!std::isfinite(ntrans[e])) return false;
    set_pose(P, nrot, ntrans);
    return true;
}

}  // namespace hm
}  // namespace tsdf
