"""np_oracle.py -- a second, NumPy restatement of the reference's hot path, written from the reference's lines
(not from tsdf_oracle.c) and structured differently (whole-array f32/f64 NumPy operations with masks instead of
per-voxel / per-corner C loops), so that an error of reading or of coding in one of the two restatements shows up
as a disagreement.  TEST INFRASTRUCTURE ONLY (tests/ and tools/make_golden.py import it; nothing shipped does).

PARITY UNPINNED vs the real reference binary, like the C oracle: the reference holds no golden vectors and cannot
be built here (ROS / PCL / Eigen / boost).  What this file adds is a second opinion, nothing more.

Reference lines restated (paths relative to /root/reference/src):
  Volume.__init__            SDF::SDF                               src/sdf.cpp:8-51 (init values :29-34, m_div_* :19-21)
  voxel_of_world             SDF::get_voxel_coordinates(world)      include/sdf_3d_reconstruction/sdf.h:143-147
  world_of_voxel             SDF::get_global_coordinates            sdf.h:153-157
  interpolate_distance       SDF::interpolate_distance              src/sdf.cpp:127-163
  update                     SDF::update                            src/sdf.cpp:224-315 (+ sdf.h:177-181, camera_tracking.cpp:40-54)
  Tracker.*                  CameraTracking                         src/camera_tracking.cpp:3-18, :59-65
  perturbed_rotations        src/camera_tracking.cpp:92-145
  accumulate                 the pixel loop + get_partial_derivative   src/camera_tracking.cpp:146-189, :246-363
  gn_update                  src/camera_tracking.cpp:191-239
  exp_map                    eigen_utils::direct_exponential_map    src/eigen_utils.cpp:40-128

Arithmetic types are the reference's: `float` members and locals are np.float32, Eigen `Vector3d` / `Matrix3d`
are np.float64; a C++ expression mixing the two is widened where C++ widens it.  Eigen 3.2 evaluation orders
(the reference pins no version; its build era is Ubuntu 12.04/14.04): fixed-size matrix products accumulate
left to right, ((a0 b0 + a1 b1) + a2 b2); dot() / norm() reduce as a0 b0 + (a1 b1 + a2 b2).
g++ without -O / -march: no fused multiply-add, so every NumPy multiply and add below is one rounding, like there.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
f64 = np.float64


EIGEN_ORDER = 32
"""32: Eigen 3.2's sequential fixed-size products ((a0 b0 + a1 b1) + a2 b2) -- the default, the reference's build era.
33: the lazy-product evaluator of Eigen >= 3.3, whose coefficient is a .sum() redux: a0 b0 + (a1 b1 + a2 b2).
(set_eigen_order; the C oracle has the same switch, orc_set_eigen_order)"""


def set_eigen_order(version):
    global EIGEN_ORDER
    EIGEN_ORDER = 33 if int(version) >= 33 else 32


def _prod3(a0, b0, a1, b1, a2, b2):
    return a0 * b0 + (a1 * b1 + a2 * b2) if EIGEN_ORDER >= 33 else (a0 * b0 + a1 * b1) + a2 * b2


def _mat3_vec(M, x, y, z):
    """Fixed-size Matrix3d * Vector3d, for arrays of vectors, in the order of EIGEN_ORDER."""
    return [_prod3(M[r, 0], x, M[r, 1], y, M[r, 2], z) for r in range(3)]


def _mat3_mat3(A, B):
    out = np.empty((3, 3), dtype=f64)
    for r in range(3):
        for c in range(3):
            out[r, c] = _prod3(A[r, 0], B[0, c], A[r, 1], B[1, c], A[r, 2], B[2, c])
    return out


def _int_cast(a):
    """C `(int) x` on x86-64 for float / double arrays: truncation toward zero; NaN and out-of-range give INT_MIN."""
    a = np.asarray(a)
    with np.errstate(invalid="ignore"):
        ok = np.isfinite(a) & (a > -2147483649.0) & (a < 2147483648.0)
        return np.where(ok, np.trunc(np.where(ok, a, 0)), -2147483648.0).astype(np.int64)


class Volume:
    """class SDF's state (sdf.h:39-56) and its constructor's values (sdf.cpp:8-34)."""

    def __init__(self, m, width, height, depth, origin, delta, epsilon):
        self.m = int(m)
        self.width, self.height, self.depth = f32(width), f32(height), f32(depth)
        self.origin = np.asarray(origin, dtype=f64)
        self.delta, self.epsilon = f32(delta), f32(epsilon)
        n = self.m ** 3
        self.m_div_height = f32(self.m) / self.height          # sdf.cpp:19-21: int / float -> float
        self.m_div_width = f32(self.m) / self.width
        self.m_div_depth = f32(self.m) / self.depth
        self.D = np.full(n, self.width + self.height + self.depth, dtype=f32)     # :29
        self.W = np.zeros(n, dtype=f32)
        self.Color_W = np.zeros(n, dtype=f32)
        self.R = np.full(n, 0.4, dtype=f32)                     # :32-34 (double literal 0.4 narrowed to float)
        self.G = np.full(n, 0.4, dtype=f32)
        self.B = np.full(n, 0.4, dtype=f32)

    # sdf.h:143-147 -- f64 world coordinate, float scale widened, minus 0.5
    def voxel_of_world(self, gx, gy, gz):
        return ((gx - self.origin[0]) * f64(self.m_div_width) - 0.5,
                (gy - self.origin[1]) * f64(self.m_div_height) - 0.5,
                (gz - self.origin[2]) * f64(self.m_div_depth) - 0.5)

    # sdf.h:153-157 -- (extent / (float) m) is a float quotient, widened for the product with (index + 0.5)
    def world_of_voxel(self, i, j, k):
        cw, ch, cd = f64(self.width / f32(self.m)), f64(self.height / f32(self.m)), f64(self.depth / f32(self.m))
        return (cw * (np.asarray(i, dtype=f64) + 0.5) + self.origin[0],
                ch * (np.asarray(j, dtype=f64) + 0.5) + self.origin[1],
                cd * (np.asarray(k, dtype=f64) + 0.5) + self.origin[2])

    def interpolate_distance(self, vox):
        """sdf.cpp:127-163 for an (n,3) array of f64 voxel coordinates -> (values f32, is_interpolated bool).

        The three nested offset loops become one pass per corner over all points; the early `return D[a_idx]`
        on an exact hit becomes a per-point `done` latch (later corners are ignored for that point, as after a
        return).  With no valid corner the reference returns 0/0 = NaN and is_interpolated stays false."""
        vox = np.asarray(vox, dtype=f64).reshape(-1, 3)
        n = len(vox)
        fi, fj, fk = vox[:, 0].astype(f32), vox[:, 1].astype(f32), vox[:, 2].astype(f32)      # :130-132
        bi, bj, bk = _int_cast(fi), _int_cast(fj), _int_cast(fk)
        m = self.m
        w_sum = np.zeros(n, dtype=f32)
        sum_d = np.zeros(n, dtype=f32)
        result = np.zeros(n, dtype=f32)
        done = np.zeros(n, dtype=bool)
        interp = np.zeros(n, dtype=bool)
        for io in (0, 1):
            for jo in (0, 1):
                for ko in (0, 1):
                    ci, cj, ck = bi + io, bj + jo, bk + ko
                    # sdf.cpp:146 fabs(int - float): the int is converted to float, the difference and the sums are float.
                    # Float, not C's double fabs, because `using namespace std;` (camera_tracking.h:11, included by
                    # sdf.h:29 ahead of sdf.cpp) makes the call resolve to the std::fabs(float) overload.
                    vol = (np.abs(ci.astype(f32) - fi) + np.abs(cj.astype(f32) - fj)) + np.abs(ck.astype(f32) - fk)
                    inside = (ci >= 0) & (cj >= 0) & (ck >= 0) & (ci < m) & (cj < m) & (ck < m)       # sdf.h:113-119
                    idx = np.where(inside, (m * m) * ci + m * cj + ck, 0)
                    take = inside & (self.W[idx] > 0) & ~done                                        # :147-149
                    interp |= take
                    hit = take & (vol.astype(f64) < 0.00001)                                         # :151 (float vs double literal)
                    result = np.where(hit, self.D[idx], result)
                    done |= hit
                    acc = take & ~hit
                    with np.errstate(divide="ignore", invalid="ignore"):
                        w = (1.0 / vol.astype(f64)).astype(f32)                                      # :154 double quotient -> float w
                    with np.errstate(invalid="ignore"):
                        w_sum = np.where(acc, w_sum + w, w_sum)
                        sum_d = np.where(acc, sum_d + w * self.D[idx], sum_d)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = np.where(done, result, sum_d / w_sum)
        return out.astype(f32), interp


class Tracker:
    """CameraTracking's pose state and step sizes (camera_tracking.cpp:3-18, :59-65)."""

    def __init__(self, vol, gauss_newton_max_iteration=20, maximum_twist_diff=0.001, v_h=1.0, w_h=0.01):
        self.K = None
        self.max_iter = int(gauss_newton_max_iteration)
        self.maximum_twist_diff = f32(maximum_twist_diff)
        self.v_h, self.w_h = f32(v_h), f32(w_h)
        self.v_h2 = f32(2) * self.v_h                           # :13  int * float -> float
        self.w_h2 = f32(2) * self.w_h
        self.v_h2_width = self.v_h2 / vol.m_div_width           # :15-17 float / float
        self.v_h2_height = self.v_h2 / vol.m_div_height
        self.v_h2_depth = self.v_h2 / vol.m_div_depth
        self.set_camera_transformation(np.array([[1.0, 0, 0], [0, 0, -1.0], [0, -1.0, 0]]), np.array([0.0, 0.0, 1.0]))

    def set_camera_transformation(self, rot, trans):            # :59-65
        self.rot = np.array(rot, dtype=f64).reshape(3, 3)
        self.rot_inv = inverse3(self.rot)
        self.trans = np.array(trans, dtype=f64).reshape(3)
        t = _mat3_vec(self.rot_inv, self.trans[0], self.trans[1], self.trans[2])
        self.rot_inv_trans = np.array([-1 * t[0], -1 * t[1], -1 * t[2]], dtype=f64)

    def perturbed_rotations(self):
        """r1p r1m r2p r2m r3p r3m = (I +- w_h [e_k]x) rot  (camera_tracking.cpp:92-145; w_h widened to double)."""
        wh = f64(self.w_h)
        out = []
        for axis in range(3):
            for s in (wh, -wh):
                Rd = np.eye(3)
                if axis == 0:
                    Rd[1, 2], Rd[2, 1] = -s, s
                elif axis == 1:
                    Rd[0, 2], Rd[2, 0] = s, -s
                else:
                    Rd[0, 1], Rd[1, 0] = -s, s
                out.append(_mat3_mat3(Rd, self.rot))
        return out


def inverse3(M):
    """Matrix3d::inverse(): cofactors over the determinant (Eigen's 3x3 path: det from the first column of cofactors)."""
    M = np.asarray(M, dtype=f64)

    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return M[i1, j1] * M[i2, j2] - M[i1, j2] * M[i2, j1]
    c00, c10, c20 = cof(0, 0), cof(1, 0), cof(2, 0)
    det = c00 * M[0, 0] + (c10 * M[1, 0] + c20 * M[2, 0])
    inv = 1.0 / det
    out = np.empty((3, 3), dtype=f64)
    for r in range(3):
        for c in range(3):
            out[r, c] = cof(c, r) * inv
    return out


def update(vol, trk, xyz, nrm, rgb, with_color=True):
    """SDF::update (sdf.cpp:224-315) over all voxels at once.  xyz / nrm: (h,w,3) float32, rgb (h,w,3) uint8.
    Returns the number of voxels rewritten."""
    m = vol.m
    h, w = xyz.shape[:2]
    idx = np.arange(m ** 3, dtype=np.int64)
    i, j, k = idx // (m * m), (idx % (m * m)) // m, idx % m                    # sdf.h:132-136
    gx, gy, gz = vol.world_of_voxel(i, j, k)                                  # the constructor's global_coords table
    c = _mat3_vec(trk.rot_inv, gx, gy, gz)                                    # camera_tracking.cpp:51-54
    pcx, pcy, pcz = c[0] + trk.rot_inv_trans[0], c[1] + trk.rot_inv_trans[1], c[2] + trk.rot_inv_trans[2]
    alive = ~(pcz < 0)                                                        # :247-249
    ij = _mat3_vec(trk.K, pcx, pcy, pcz)                                      # camera_tracking.cpp:40-47
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = ij[0] / ij[2], ij[1] / ij[2]
    iu, iv = _int_cast(u), _int_cast(v)                                       # :251-252
    alive &= (iu >= 0) & (iv >= 0) & (iu < w) & (iv < h)                      # :254 (unsigned compare + explicit < 0)
    cu, cv = np.where(alive, iu, 0), np.where(alive, iv, 0)
    P = xyz[cv, cu].astype(f32)                                               # at(col, row)
    N = nrm[cv, cu].astype(f32)
    alive &= ~(np.isnan(P[:, 0]) | np.isnan(P[:, 1]) | np.isnan(N[:, 0]) | np.isnan(N[:, 1]) | np.isnan(N[:, 2]))   # :260
    # sdf.h:177-181: diff = camera_point_img - camera_point; diff.dot(normal) = d0 n0 + (d1 n1 + d2 n2)
    Pd, Nd = P.astype(f64), N.astype(f64)
    dx, dy, dz = Pd[:, 0] - pcx, Pd[:, 1] - pcy, Pd[:, 2] - pcz
    p2p = dx * Nd[:, 0] + (dy * Nd[:, 1] + dz * Nd[:, 2])
    d_new = p2p.astype(f32)                                                   # :274
    band = alive & (d_new >= vol.epsilon) & (d_new <= vol.delta)              # :277
    w_new = np.ones(len(idx), dtype=f32)
    a = (d_new - vol.epsilon).astype(f32)                                     # float - float
    arg = (-0.5 * a.astype(f64)) * a.astype(f64)                              # -0.5 * (float) * (float), left to right, in double
    bi = np.nonzero(band)[0]
    w_new[bi] = np.array([math.exp(x) for x in arg[bi]], dtype=f64).astype(f32)   # libm exp (the C library the reference links)
    alive &= ~(d_new > vol.delta)                                             # :280-283
    d_new = np.where(d_new < -vol.delta, -vol.delta, d_new).astype(f32)       # :285-287
    sel = np.nonzero(alive)[0]
    w_old = vol.W[sel]
    wn, dn = w_new[sel], d_new[sel]
    W2 = w_old + wn                                                           # :289-290
    vol.D[sel] = (w_old * vol.D[sel] + wn * dn) / W2                          # :292
    vol.W[sel] = W2
    if with_color:
        n0, n1, n2 = Nd[sel, 0], Nd[sel, 1], Nd[sel, 2]
        cosine = np.abs(0.0 * n0 + (0.0 * n1 + 1.0 * n2)) / np.sqrt(n0 * n0 + (n1 * n1 + n2 * n2))   # :294
        cw_old = vol.Color_W[sel]
        wc = (wn.astype(f64) * cosine).astype(f32)                            # :298 float * double -> double -> float
        CW2 = cw_old + wc
        col = rgb[cv[sel], cu[sel]].astype(f32)                               # uint8 -> int -> float in `w_new * point.r`
        vol.R[sel] = (cw_old * vol.R[sel] + wc * col[:, 0]) / CW2
        vol.G[sel] = (cw_old * vol.G[sel] + wc * col[:, 1]) / CW2
        vol.B[sel] = (cw_old * vol.B[sel] + wc * col[:, 2]) / CW2
        vol.Color_W[sel] = CW2
    return int(len(sel))


def accumulate(vol, trk, xyz, stale_carry=True, stride=3):
    """One pass of the pixel loop of estimate_new_position (camera_tracking.cpp:146-189) with ONE OpenMP thread:
    columns outer, rows inner, stride 3 (:162-163).  Returns (A 6x6, b 6, stats).

    The thread-local SDF_derivative / int_dist / is_interpolated live across pixels (:156-159): a pixel whose voxel
    lies outside the grid returns from get_partial_derivative before anything is written (:261-268), so the test
    `if (!is_interpolated) continue` (:176) sees the previous pixel's flag and, when that was a success, the previous
    J and r are added once more.  stale_carry=False restates the loop with the flag reset per pixel."""
    h, w = xyz.shape[:2]
    cols, rows = np.arange(0, w, stride), np.arange(0, h, stride)
    pts = xyz[rows[None, :], cols[:, None]].astype(f32).reshape(-1, 3)       # visiting order
    n = len(pts)
    nan = np.isnan(pts[:, 0]) | np.isnan(pts[:, 1]) | np.isnan(pts[:, 2])    # :168
    px, py, pz = pts[:, 0].astype(f64), pts[:, 1].astype(f64), pts[:, 2].astype(f64)
    wpt = _mat3_vec(trk.rot, px, py, pz)                                       # :55-58
    v0 = vol.voxel_of_world(wpt[0] + trk.trans[0], wpt[1] + trk.trans[1], wpt[2] + trk.trans[2])
    with np.errstate(invalid="ignore"):
        oog = (v0[0] < 0) | (v0[1] < 0) | (v0[2] < 0) | (v0[0] >= vol.m) | (v0[1] >= vol.m) | (v0[2] >= vol.m)   # :261-268
    # the 13 look-ups of every sample, evaluated for all samples (the reference stops at the first failure;
    # a failed sample contributes nothing either way, so evaluating the rest changes no result)
    V0 = np.stack(v0, 1)
    vals = np.zeros((13, n), dtype=f32)
    oks = np.zeros((13, n), dtype=bool)
    vals[0], oks[0] = vol.interpolate_distance(V0)                           # :269
    vh = f64(trk.v_h)
    for a in range(3):                                                       # :273-316
        for s, slot in ((vh, 1 + 2 * a), (-vh, 2 + 2 * a)):
            Vp = V0.copy()
            Vp[:, a] = Vp[:, a] + s
            vals[slot], oks[slot] = vol.interpolate_distance(Vp)
    for q, Rq in enumerate(trk.perturbed_rotations()):                       # :318-361: r_k+- * camera_point + trans
        wq = _mat3_vec(Rq, px, py, pz)
        vq = vol.voxel_of_world(wq[0] + trk.trans[0], wq[1] + trk.trans[1], wq[2] + trk.trans[2])
        vals[7 + q], oks[7 + q] = vol.interpolate_distance(np.stack(vq, 1))
    good = oks.all(axis=0)
    J = np.zeros((n, 6), dtype=f64)
    for a, hstep in enumerate((trk.v_h2_width, trk.v_h2_height, trk.v_h2_depth)):
        J[:, a] = ((vals[1 + 2 * a] - vals[2 + 2 * a]) / hstep).astype(f64)   # float - float, / float, widened on assignment
    den = f32(2) * trk.w_h                                                   # (2 * (w_h)): int * float
    for a in range(3):
        J[:, 3 + a] = ((vals[7 + 2 * a] - vals[8 + 2 * a]) / den).astype(f64)
    r = vals[0].astype(f64)
    A = np.zeros((6, 6), dtype=f64)
    b = np.zeros(6, dtype=f64)
    st = {"n_samples": n, "n_nan": 0, "n_oog": 0, "n_fail": 0, "n_ok": 0, "n_terms": 0}
    flag, Jc, rc = False, np.zeros(6), 0.0                                    # thread-locals, reset per pass (:156-159)
    for s in range(n):
        if nan[s]:
            st["n_nan"] += 1
            continue
        if not stale_carry:
            flag = False
        if oog[s]:
            st["n_oog"] += 1                                                 # returns before touching anything
        elif good[s]:
            st["n_ok"] += 1
            flag, Jc, rc = True, J[s], r[s]
        else:
            st["n_fail"] += 1
            flag = False
        if not flag:
            continue
        A = A + np.outer(Jc, Jc)                                             # :181
        b = b + rc * Jc                                                      # :182
        st["n_terms"] += 1
    return A, b, st


def exp_map(twist, delta_t=1.0):
    """eigen_utils::direct_exponential_map (eigen_utils.cpp:85-128) with UThetaToAffine3d (:61-83) and the
    f_sinc / f_mcosc / f_msinc guards (:40-59).  Returns (R 3x3, t 3)."""
    v = np.asarray(twist, dtype=f64) * f64(delta_t)
    u = v[3:6]
    theta = math.sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2])
    si, co = math.sin(theta), math.cos(theta)
    sinc = 1.0 if abs(theta) < 1.0e-8 else si / theta
    mcosc = 0.5 if abs(theta) < 2.5e-4 else (1.0 - co) / theta / theta
    msinc = (1. / 6.0) if abs(theta) < 2.5e-4 else (1.0 - si / theta) / theta / theta
    R = np.array([[co + mcosc * u[0] * u[0], -sinc * u[2] + mcosc * u[0] * u[1], sinc * u[1] + mcosc * u[0] * u[2]],
                  [sinc * u[2] + mcosc * u[1] * u[0], co + mcosc * u[1] * u[1], -sinc * u[0] + mcosc * u[1] * u[2]],
                  [-sinc * u[1] + mcosc * u[2] * u[0], sinc * u[0] + mcosc * u[2] * u[1], co + mcosc * u[2] * u[2]]])
    t = np.array([
        v[0] * (sinc + u[0] * u[0] * msinc) + v[1] * (u[0] * u[1] * msinc - u[2] * mcosc) + v[2] * (u[0] * u[2] * msinc + u[1] * mcosc),
        v[0] * (u[0] * u[1] * msinc + u[2] * mcosc) + v[1] * (sinc + u[1] * u[1] * msinc) + v[2] * (u[1] * u[2] * msinc - u[0] * mcosc),
        v[0] * (u[0] * u[2] * msinc - u[1] * mcosc) + v[1] * (u[1] * u[2] * msinc + u[0] * mcosc) + v[2] * (sinc + u[2] * u[2] * msinc)])
    return R, t


def gn_update(trk, A, b):
    """camera_tracking.cpp:191-239: twist = A^-1 b (LAPACK's partial-pivot LU here, Eigen's in the reference: the same
    algorithm in another operation order, so the twist agrees to rounding, not bit for bit), exponential map, the
    SIGNED stop test on all six components (:216-224), then rot <- R^T rot, trans <- trans - R^T t (:237-239; the
    update is applied on the stopping pass too).  Returns (twist, stop)."""
    twist = np.linalg.inv(np.asarray(A, dtype=f64)) @ np.asarray(b, dtype=f64)
    R, t = exp_map(twist, 1.0)
    thr = f64(trk.maximum_twist_diff)
    stop = bool(np.all(twist < thr))
    Rt = R.T.copy()                                   # aff.rotation(): the linear block (already orthogonal)
    rot = _mat3_mat3(Rt, trk.rot)
    rt = _mat3_vec(Rt, t[0], t[1], t[2])
    trans = np.array([trk.trans[0] - rt[0], trk.trans[1] - rt[1], trk.trans[2] - rt[2]])
    trk.set_camera_transformation(rot, trans)
    return twist, stop


def estimate_new_position(vol, trk, xyz, stale_carry=True):
    """camera_tracking.cpp:66-245 with one thread.  Returns {iterations, stopped, last_twist, n_terms_last}."""
    stop, g, twist, st = False, 0, np.zeros(6), {"n_terms": 0}
    while g < trk.max_iter and not stop:
        A, b, st = accumulate(vol, trk, xyz, stale_carry)
        twist, stop = gn_update(trk, A, b)
        g += 1
    return {"iterations": g, "stopped": int(stop), "last_twist": twist, "n_terms_last": st["n_terms"]}
