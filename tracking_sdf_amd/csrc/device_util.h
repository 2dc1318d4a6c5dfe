// device_util.h -- small device helpers shared by the integration and tracker kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

namespace tsdf {

// (int)float / (int)double as x86-64 cvttss2si / cvttsd2si: out-of-range and NaN give INT_MIN
// (the reference relies on this at sdf.cpp:143-145 and :251-252; v_cvt_i32_f32 would saturate).
__device__ __forceinline__ int trunc_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// Eigen 3.2 coefficient-based 3x3 * 3: ((a0*b0 + a1*b1) + a2*b2), see DESIGN.md "evaluation orders".
__device__ __forceinline__ double row3(const double* M, double x, double y, double z) {
    return (M[0] * x + M[1] * y) + M[2] * z;
}

__device__ __forceinline__ bool is_nan(float f) { return f != f; }

// sdf.cpp:294  cosine = fabs(cam_vect.dot(n)) / n.norm(), cam_vect = (0,0,1), in the reference's f64
// evaluation order (Eigen redux: a0*b0 + (a1*b1 + a2*b2)).
__device__ __forceinline__ double pixel_cosine(float nx, float ny, float nz) {
    const double x = (double)nx, y = (double)ny, z = (double)nz;
    return fabs(0.0 * x + (0.0 * y + 1.0 * z)) / sqrt(x * x + (y * y + z * z));
}

// The same value for normals of ordinary size, without the wrappers hipcc puts around sqrt() and '/': its correctly
// rounded cores as they stand in this build's ISA (v_rsq_f64 + the coupled iteration with two residual corrections;
// v_rcp_f64 + two Newton steps, quotient, one residual correction), minus the range scaling (v_div_scale / v_ldexp),
// which only acts outside 2^+-767 / 2^+-1022, and minus the special-case selects (v_div_fixup, the 0 / inf pass-through).
// With x, y finite, fabs(0*x + (0*y + z)) is |z|.  Valid for 2^-200 <= n2 <= 2^200 (the caller tests n2, which also
// rules out NaN and infinite components); bit-identical to pixel_cosine() there -- the fuzz and parity tests compare
// the colour weights of band voxels with the oracle's libm sqrt and division.
__device__ __forceinline__ double pixel_cosine_core(double z, double n2) {
    const double y = __builtin_amdgcn_rsq(n2);
    double g = n2 * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, n2);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, n2);
    g = __builtin_fma(d, h, g);                               // sqrt(n2)
    const double a = __builtin_fabs(z);
    double q = __builtin_amdgcn_rcp(g);
    double e = __builtin_fma(-g, q, 1.0);
    q = __builtin_fma(q, e, q);
    e = __builtin_fma(-g, q, 1.0);
    q = __builtin_fma(q, e, q);
    const double t = a * q;
    const double res = __builtin_fma(-g, t, a);
    return __builtin_fma(res, q, t);
}

}  // namespace tsdf
