"""Known-answer tests that pin the CPU oracle to the reference's source lines.

The reference ships no tests or golden vectors, so every
expected value here is derived by hand from the cited lines of
/root/reference/src (KAT-1..8 of SURVEY.md section 8c).
"""
import math

import numpy as np
import pytest

import oracle as orc


def small_sdf(m=8, ext=(8.0, 8.0, 8.0), origin=(0.0, 0.0, 0.0), **kw):
    return orc.SDF(m, ext[0], ext[1], ext[2], origin, kw.get("delta", 0.3), kw.get("epsilon", 0.025))


# ---- KAT-1: index maps, sdf.h:113-136
def test_kat1_index_roundtrip():
    s = small_sdf(m=5)
    for idx in range(5 ** 3):
        v = s.get_voxel_coordinates_idx(idx)
        assert s.get_array_index(v) == idx
        assert idx == 25 * v[0] + 5 * v[1] + v[2]          # k fastest, sdf.h:120
    for bad in ([-1, 0, 0], [0, -1, 0], [0, 0, -1], [5, 0, 0], [0, 5, 0], [0, 0, 5]):
        assert s.get_array_index(bad) == -1


# ---- constructor values, sdf.cpp:19-34
def test_ctor_init_values():
    s = orc.SDF(16, 6.0, 6.0, 3.5, (-3, -3, -0.5), 0.3, 0.025)
    assert np.all(s.D == np.float32(15.5)) and np.all(s.W == 0) and np.all(s.Color_W == 0)
    assert np.all(s.R == np.float32(0.4)) and np.all(s.G == np.float32(0.4)) and np.all(s.B == np.float32(0.4))
    assert s.c.m_div_depth == np.float32(16) / np.float32(3.5)      # float quotient
    assert s.c.m_div_width == np.float32(16) / np.float32(6.0)


# ---- KAT-2: inverse-L1 interpolation, sdf.cpp:127-163
def test_kat2_interpolation_is_not_trilinear():
    m = 8
    s = small_sdf(m)
    D = s.D.reshape(m, m, m)
    s.W[:] = 1.0
    D[:] = np.arange(m, dtype=np.float32)[None, :, None]            # D = j
    val, ok = s.interpolate_distance([3.5, 2.0, 4.0])
    assert ok
    assert val == pytest.approx(2.0 + 2.0 / 7.0, abs=2e-6)          # trilinear would give 2.0
    # exact corner hit returns that corner's D (early return, sdf.cpp:151-153)
    val, ok = s.interpolate_distance([3.0, 5.0, 1.0])
    assert ok and val == 5.0
    # truncation toward zero: -0.5 uses base corner 0, not -1 (sdf.cpp:143-145)
    D[:] = 1.0
    D[0, 0, 0] = 9.0
    s.W[:] = 0.0
    s.W.reshape(m, m, m)[0, 0, 0] = 1.0
    val, ok = s.interpolate_distance([-0.5, -0.25, -0.25])
    assert ok and val == 9.0                                         # only valid corner
    # no valid corner: is_interpolated false, value 0/0 = NaN
    val, ok = s.interpolate_distance([4.2, 4.2, 4.2])
    assert (not ok) and math.isnan(val)
    # out of grid entirely
    val, ok = s.interpolate_distance([-3.0, 100.0, 2.0])
    assert (not ok) and math.isnan(val)
    # exact hit on a corner with W == 0 falls through to the weighted mean of the others
    s.W[:] = 1.0
    s.W.reshape(m, m, m)[2, 2, 2] = 0.0
    D[:] = 3.0
    D[2, 2, 2] = 100.0
    val, ok = s.interpolate_distance([2.0, 2.0, 2.0])
    assert ok and val == pytest.approx(3.0, abs=1e-6)


def test_kat2_weights_are_float32():
    """w = 1/L1 and the sums are float (sdf.cpp:133-137,154-156): compare with a float32 re-derivation."""
    m = 8
    s = small_sdf(m)
    rng = np.random.default_rng(3)
    s.W[:] = 1.0
    s.D[:] = rng.standard_normal(m ** 3).astype(np.float32)
    D = s.D.reshape(m, m, m)
    for _ in range(200):
        v = rng.uniform(0.0, m - 1.001, size=3)
        f = v.astype(np.float32)
        base = np.trunc(f).astype(np.int32)
        wsum = np.float32(0)
        dsum = np.float32(0)
        for io in (0, 1):
            for jo in (0, 1):
                for ko in (0, 1):
                    c = base + np.array([io, jo, ko], dtype=np.int32)
                    vol = np.float32(np.float32(abs(np.float32(c[0]) - f[0])) + np.float32(abs(np.float32(c[1]) - f[1])))
                    vol = np.float32(vol + np.float32(abs(np.float32(c[2]) - f[2])))
                    w = np.float32(np.float32(1.0) / vol)
                    wsum = np.float32(wsum + w)
                    dsum = np.float32(dsum + np.float32(w * D[c[0], c[1], c[2]]))
        want = np.float32(dsum / wsum)
        got, ok = s.interpolate_distance(v)
        assert ok and np.float32(got) == want


# ---- KAT-3: coordinate maps, sdf.h:143-157
def test_kat3_world_voxel_maps():
    s = orc.SDF(256, 6.0, 6.0, 3.5, (-3, -3, -0.5), 0.3, 0.025)
    g = s.get_global_coordinates([0, 0, 0])
    f32 = np.float32
    want = np.array([float(f32(6.0) / f32(256)) * 0.5 - 3.0,
                     float(f32(6.0) / f32(256)) * 0.5 - 3.0,
                     float(f32(3.5) / f32(256)) * 0.5 - 0.5])
    assert np.array_equal(g, want)
    # voxel centres sit at integer voxel coordinates
    for vox in ([0, 0, 0], [17, 200, 255], [255, 255, 255]):
        v = s.get_voxel_coordinates(s.get_global_coordinates(vox))
        assert np.allclose(v, vox, atol=1e-5)
    # m_div_* are floats: (g - origin) * float(m/extent) - 0.5
    v = s.get_voxel_coordinates([0.1234, -1.5, 1.0])
    want = np.array([(0.1234 + 3.0) * float(f32(256) / f32(6.0)) - 0.5,
                     (-1.5 + 3.0) * float(f32(256) / f32(6.0)) - 0.5,
                     (1.0 + 0.5) * float(f32(256) / f32(3.5)) - 0.5])
    assert np.array_equal(v, want)


# ---- KAT-4: exponential map, eigen_utils.cpp:43-128
def test_kat4_exp_map():
    T = orc.direct_exponential_map([1, 2, 3, 0, 0, 0])
    assert np.array_equal(T[:, :3], np.eye(3)) and np.allclose(T[:, 3], [1, 2, 3], atol=1e-15)
    T = orc.direct_exponential_map([1, 0, 0, 0, 0, math.pi / 2])
    Rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    assert np.allclose(T[:, :3], Rz, atol=1e-15)
    assert np.allclose(T[:, 3], [2 / math.pi, 2 / math.pi, 0], atol=1e-15)
    # generic twist vs matrix exponential of the 4x4 twist matrix
    from scipy.linalg import expm
    xi = np.array([0.03, -0.02, 0.05, 0.1, -0.2, 0.15])
    w = xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = xi[:3]
    assert np.allclose(orc.direct_exponential_map(xi), expm(M)[:3, :], atol=1e-14)
    # small-angle guards (eigen_utils.cpp:40-59): theta < 2.5e-4 uses 1/2 and 1/6
    xi = np.array([1.0, 0, 0, 0, 0, 1e-4])
    T = orc.direct_exponential_map(xi)
    assert T[0, 0] == math.cos(1e-4) and T[1, 0] == math.sin(1e-4) / 1e-4 * 1e-4
    assert T[1, 3] == 1.0 * (1e-4 * 0.5)            # v0 * (u2 * mcosc) with the guarded mcosc = 0.5


# ---- Eigen restatements
def test_inverse3_and_inverse6():
    rot0 = np.array([[1, 0, 0], [0, 0, -1], [0, -1, 0.0]])          # camera_tracking.cpp:7, det = -1
    assert np.array_equal(orc.inverse3(rot0), rot0.T)
    rng = np.random.default_rng(0)
    for _ in range(20):
        M = rng.standard_normal((3, 3))
        assert np.allclose(orc.inverse3(M) @ M, np.eye(3), atol=1e-10)
        J = rng.standard_normal((50, 6))
        A = J.T @ J
        Ai, ok = orc.inverse6(A)
        assert ok and np.allclose(Ai, np.linalg.inv(A), rtol=1e-9, atol=1e-12)


# ---- KAT-8 + pose algebra: camera_tracking.cpp:3-18, 59-65
def test_kat8_ctor_arg_order_and_initial_pose():
    s = orc.SDF(256, 6.0, 6.0, 3.5, (-3, -3, -0.5), 0.3, 0.025)
    t = orc.CameraTracking(s, 20, 0.001, 1.0, 0.01)                  # call site sdf_reconstruction.cpp:88
    c = t.c
    assert c.gauss_newton_max_iteration == 20
    assert c.maximum_twist_diff == np.float32(0.001)
    assert c.v_h == 1.0 and c.w_h == np.float32(0.01)
    assert c.v_h2_width == np.float32(2.0) / (np.float32(256) / np.float32(6.0))
    assert c.v_h2_depth == np.float32(2.0) / (np.float32(256) / np.float32(3.5))
    assert np.array_equal(t.trans, [0, 0, 1])
    assert np.array_equal(t.rot, [[1, 0, 0], [0, 0, -1], [0, -1, 0]])
    assert np.array_equal(t.rot_inv, t.rot.T)
    assert np.array_equal(t.rot_inv_trans, [0, 1, 0])                # -(rot_inv * (0,0,1)) = (0,1,0)


# ---- KAT-5: SDF::update on a fronto-parallel plane, sdf.cpp:224-315
def _plane_frame(w, h, z0, K):
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    xyz = np.zeros((h, w, 3), dtype=np.float32)
    xyz[..., 2] = z0
    xyz[..., 0] = (u - K[0, 2]) / K[0, 0] * z0
    xyz[..., 1] = (v - K[1, 2]) / K[1, 1] * z0
    nrm = np.zeros((h, w, 3), dtype=np.float32)
    nrm[..., 2] = -1.0
    rgb = np.full((h, w, 3), 200, dtype=np.uint8)
    return xyz, nrm, rgb


def test_kat5_update_plane():
    m = 32
    s = orc.SDF(m, 3.2, 3.2, 3.2, (-1.6, -1.6, 0.0), 0.3, 0.025)    # voxel = 0.1 m, camera at the origin
    t = orc.CameraTracking(s)
    t.set_camera_transformation(np.eye(3), np.zeros(3))
    K = np.array([[50.0, 0, 31.5], [0, 50.0, 23.5], [0, 0, 1]])
    t.set_K(K)
    z0 = 1.5
    xyz, nrm, rgb = _plane_frame(64, 48, z0, K)
    cloud = orc.Cloud(xyz, nrm, rgb)
    n = s.update(t, cloud, with_color=True)
    assert n > 0
    D = s.D.reshape(m, m, m)
    W = s.W.reshape(m, m, m)
    # voxels nearest the optical axis: x,y index 16 -> centre (0.05, 0.05, z)
    for k in range(m):
        z = float(np.float32(3.2) / np.float32(m)) * (k + 0.5)
        d = np.float32(z - z0)                                       # (P - pc) . N with N = (0,0,-1)
        d0 = np.float32(3.2) + np.float32(3.2) + np.float32(3.2)     # init value, sdf.cpp:29
        x = float(np.float32(3.2) / np.float32(m)) * 16.5 - 1.6
        u, v = 50.0 * x / z + 31.5, 50.0 * x / z + 23.5
        if not (-1 < u < 64 and -1 < v < 48):
            assert W[16, 16, k] == 0 and D[16, 16, k] == d0          # projects outside the image
        elif d > np.float32(0.3):
            assert W[16, 16, k] == 0 and D[16, 16, k] == d0          # behind the surface by > delta: skipped
        else:
            dd = max(d, np.float32(-0.3))
            wexp = np.float32(1.0)
            if np.float32(0.025) <= d <= np.float32(0.3):
                a = np.float32(d - np.float32(0.025))
                wexp = np.float32(math.exp(-0.5 * float(a) * float(a)))
            assert W[16, 16, k] == wexp
            assert D[16, 16, k] == pytest.approx(float(dd), abs=1e-6)     # first update ignores the init value
            assert s.Color_W.reshape(m, m, m)[16, 16, k] == wexp     # cosine = |n.z|/|n| = 1
            assert s.R.reshape(m, m, m)[16, 16, k] == pytest.approx(200.0, rel=1e-6)
    # second identical update: W doubles, D unchanged (weighted mean of equal values)
    D1 = s.D.copy(); W1 = s.W.copy()
    s.update(t, cloud, with_color=True)
    upd = W1 > 0
    assert np.allclose(s.W[upd], 2 * W1[upd], rtol=1e-6)
    assert np.allclose(s.D[upd], D1[upd], atol=1e-6)
    # behind-camera voxels never touched; K missing -> -1 (reference exit(0), sdf.cpp:227-230)
    t2 = orc.CameraTracking(s)
    assert s.update(t2, cloud) == -1


def test_update_pixel_truncation_and_nan_rules():
    """(int)u truncation: u in (-1,0) lands on column 0 (sdf.cpp:251-256); NaN x/y/normal skipped (:260)."""
    m = 16
    s = orc.SDF(m, 1.6, 1.6, 1.6, (-0.8, -0.8, 0.2), 0.3, 0.025)
    t = orc.CameraTracking(s)
    t.set_camera_transformation(np.eye(3), np.zeros(3))
    K = np.array([[20.0, 0, 7.5], [0, 20.0, 5.5], [0, 0, 1]])
    t.set_K(K)
    xyz, nrm, rgb = _plane_frame(16, 12, 1.0, K)
    xyz[5, 3, 0] = np.nan          # NaN x -> skipped
    nrm[6, 4, 1] = np.nan          # NaN normal -> skipped
    xyz[7, 5, 2] = np.nan          # NaN z alone is NOT tested by the reference: the voxel is still updated
    cloud = orc.Cloud(xyz, nrm, rgb)
    s.update(t, cloud, with_color=False)
    W = s.W.reshape(m, m, m)
    touched = np.zeros((12, 16), dtype=bool)
    nanD = 0
    for idx in range(m ** 3):
        v = s.get_voxel_coordinates_idx(idx)
        g = s.get_global_coordinates(v)
        if g[2] < 0:
            continue
        u = (K @ g)[0] / g[2]
        vv = (K @ g)[1] / g[2]
        iu, iv = int(u), int(vv)          # python int() truncates toward zero like (int)
        inside = (-1 < u < 16) and (-1 < vv < 12)
        if W[tuple(v)] > 0:
            assert inside and 0 <= iu < 16 and 0 <= iv < 12
            assert (iv, iu) not in ((5, 3), (6, 4))
            touched[iv, iu] = True
            if (iv, iu) == (7, 5):
                nanD += 1
    assert touched[:, 0].any() or touched[0, :].any()
    assert nanD > 0 and np.isnan(s.D).sum() == nanD


# ---- KAT-6: signed stop rule, camera_tracking.cpp:216-224
def test_kat6_signed_stop_rule():
    s = small_sdf(8)
    t = orc.CameraTracking(s, 20, 0.001, 1.0, 0.01)
    A = np.eye(6)
    stop, tw = t.gn_update(A, -np.ones(6))          # twist = (-1,...,-1): every component < 0.001
    assert stop and np.allclose(tw, -1)
    stop, tw = t.gn_update(A, np.array([0, 0, 0, 0, 0, 0.002]))
    assert not stop
    stop, tw = t.gn_update(A, np.full(6, 0.0009))
    assert stop
    # float threshold widened: 0.001f = 0.001000000047..., so 0.00100000001 still stops
    stop, tw = t.gn_update(A, np.full(6, 0.00100000001))
    assert stop


def test_gn_update_pose_composition():
    """rot <- R^T rot ; trans <- trans - R^T t  with [R|t] = exp(+twist)  (camera_tracking.cpp:237-239)."""
    s = small_sdf(8)
    t = orc.CameraTracking(s)
    rot0, trans0 = t.rot.copy(), t.trans.copy()
    tw = np.array([0.01, -0.02, 0.03, 0.02, 0.01, -0.03])
    stop, got = t.gn_update(np.eye(6), tw)
    T = orc.direct_exponential_map(tw)
    R, tt = T[:, :3], T[:, 3]
    assert np.allclose(t.rot, R.T @ rot0, atol=1e-15)
    assert np.allclose(t.trans, trans0 - R.T @ tt, atol=1e-15)
    assert np.allclose(t.rot_inv @ t.rot, np.eye(3), atol=1e-14)
    assert np.allclose(t.rot_inv_trans, -(t.rot_inv @ t.trans), atol=1e-15)


# ---- Jacobian structure, camera_tracking.cpp:246-363
def test_partial_derivative_on_analytic_sphere():
    m = 64
    s = orc.SDF(m, 3.2, 3.2, 3.2, (-1.6, -1.6, -1.6), 0.3, 0.025)
    s.create_circle(1.0, 0.0, 0.0, 0.0)                               # D = |x| - 1, W = 1
    t = orc.CameraTracking(s)
    t.set_camera_transformation(np.eye(3), np.zeros(3))
    p = np.array([0.7, 0.5, 0.4])
    ing, J, ok, r = t.get_partial_derivative(s, p)
    assert ing and ok
    n = p / np.linalg.norm(p)
    assert r == pytest.approx(np.linalg.norm(p) - 1.0, abs=0.02)
    assert np.allclose(J[:3], n, atol=0.05)                           # translation part ~ grad(SDF)
    assert np.allclose(J[3:], np.cross(p, n), atol=0.05)              # rotation about world axes: (w x p).n = w.(p x n) = 0 for a centred sphere
    # out of grid: outputs untouched, flag untouched (camera_tracking.cpp:261-268)
    ing, J2, ok2, r2 = t.get_partial_derivative(s, [5.0, 0, 0], J=np.full(6, 7.0), is_interpolated=True, sdf_val=3.0)
    assert (not ing) and ok2 and r2 == 3.0 and np.all(J2 == 7.0)


# ---- KAT-7: stale carry, camera_tracking.cpp:156-159,176-182,261-268
def _stale_fixture():
    m = 32
    s = orc.SDF(m, 3.2, 3.2, 3.2, (0, 0, 0), 0.3, 0.025)
    s.create_circle(1.0, 1.0, 1.0, 1.0)
    s.W.reshape(m, m, m)[:8, :, :] = 0.0                              # x < 0.8 m has no data
    t = orc.CameraTracking(s)
    t.set_camera_transformation(np.eye(3), np.zeros(3))
    OK, OOG, FAIL = (1.6, 1.6, 1.6), (-1.0, 1.6, 1.6), (0.4, 1.6, 1.6)
    return s, t, OK, OOG, FAIL


def _column_cloud(points):
    """width 1, height 3*len-2: sampled rows 0,3,6,... hold `points` (None = NaN pixel)."""
    h = 3 * len(points) - 2
    xyz = np.full((h, 1, 3), 0.123, dtype=np.float32)
    for n, p in enumerate(points):
        xyz[3 * n, 0] = (np.nan, np.nan, np.nan) if p is None else p
    return orc.Cloud(xyz)


def test_kat7_stale_carry():
    s, t, OK, OOG, FAIL = _stale_fixture()
    A1, b1, st1 = t.accumulate(s, _column_cloud([OK]))
    assert st1["n_ok"] == 1 and st1["n_terms"] == 1
    ing, J, ok, r = t.get_partial_derivative(s, OK)
    assert np.allclose(A1, np.outer(J, J), rtol=1e-15) and np.allclose(b1, r * J, rtol=1e-15)
    # [OK, OOG, OOG, FAIL, OOG] => A = 3 JJ^T
    A, b, st = t.accumulate(s, _column_cloud([OK, OOG, OOG, FAIL, OOG]))
    assert st == {"n_samples": 5, "n_nan": 0, "n_oog": 3, "n_fail": 1, "n_ok": 1, "n_terms": 3}
    assert np.allclose(A, 3 * A1, rtol=1e-15) and np.allclose(b, 3 * b1, rtol=1e-15)
    # NaN pixels in between are transparent
    A, b, st = t.accumulate(s, _column_cloud([OK, None, OOG, None, OOG, FAIL, OOG]))
    assert st["n_terms"] == 3 and np.allclose(A, 3 * A1, rtol=1e-15)
    # leading OOG before any success adds nothing; state resets every call (every GN iteration)
    A, b, st = t.accumulate(s, _column_cloud([OOG, OOG, OK, OOG]))
    assert st["n_terms"] == 2 and np.allclose(A, 2 * A1, rtol=1e-15)
    # flag off: one term per successful pixel
    A, b, st = t.accumulate(s, _column_cloud([OK, OOG, OOG, FAIL, OOG]), stale_carry=False)
    assert st["n_terms"] == 1 and np.allclose(A, A1, rtol=1e-15)


def test_stale_carry_sampling_order_is_column_major():
    """i (columns) outer, j (rows) inner, stride 3 (camera_tracking.cpp:162-163): the carry runs down a column."""
    s, t, OK, OOG, FAIL = _stale_fixture()
    xyz = np.full((4, 4, 3), np.nan, dtype=np.float32)
    xyz[0, 0] = OK       # sample (col 0,row 0)
    xyz[3, 0] = FAIL     # (col 0,row 3)  next in column-major order -> ends the run
    xyz[0, 3] = OOG      # (col 3,row 0)
    xyz[3, 3] = OOG      # (col 3,row 3)
    A, b, st = t.accumulate(s, orc.Cloud(xyz))
    assert st["n_terms"] == 1          # row-major order would have given 2 (OK, OOG, FAIL, OOG)
    xyz[3, 0] = OOG
    xyz[0, 3] = FAIL
    A, b, st = t.accumulate(s, orc.Cloud(xyz))
    assert st["n_terms"] == 2          # OK, OOG, FAIL, OOG in column-major order


def test_ownership_partition_sums_to_whole():
    """Slab ownership (multi-GPU sharding) partitions the terms exactly, stale re-adds included."""
    s, t, OK, OOG, FAIL = _stale_fixture()
    OK2 = (2.45, 1.2, 1.3)
    cloud = _column_cloud([OK, OOG, OK2, OOG, OOG, FAIL, OK, OK2, OOG])
    A, b, st = t.accumulate(s, cloud)
    parts = [t.accumulate(s, cloud, own_x0=x0, own_x1=x1) for x0, x1 in ((0, 16), (16, 20), (20, 32))]
    assert sum(p[2]["n_terms"] for p in parts) == st["n_terms"] == 8
    assert np.allclose(sum(p[0] for p in parts), A, rtol=1e-14)
    assert np.allclose(sum(p[1] for p in parts), b, rtol=1e-14)
    assert parts[0][2]["n_terms"] == 3 and parts[1][2]["n_terms"] == 0 and parts[2][2]["n_terms"] == 5


# ---- estimate_new_position: converges on an analytic sphere from a perturbed pose
def test_estimate_new_position_recovers_small_offset():
    m = 64
    s = orc.SDF(m, 3.2, 3.2, 3.2, (-1.6, -1.6, -1.6), 0.3, 0.025)
    # union of three off-centre spheres: constrains all six degrees of freedom
    g = np.stack(np.meshgrid(*[(np.arange(m) + 0.5) * (3.2 / m) - 1.6] * 3, indexing="ij"), -1)
    d = np.minimum.reduce([np.linalg.norm(g - c, axis=-1) - r for c, r in
                           (((0.3, 0.2, 0.1), 0.5), ((-0.5, 0.4, -0.2), 0.35), ((0.1, -0.6, 0.3), 0.3))])
    s.D[:] = d.astype(np.float32).reshape(-1)
    s.W[:] = 1.0
    # camera at (0,0,-1.5) looking along +z, proper rotation here
    true_R, true_t = np.eye(3), np.array([0.0, 0.0, -1.5])
    K = np.array([[60.0, 0, 31.5], [0, 60.0, 23.5], [0, 0, 1]])
    w, h = 64, 48
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    dirs = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u, dtype=float)], -1)
    # ray-march the union of spheres in world coordinates
    tt = np.zeros((h, w))
    hit = np.zeros((h, w), dtype=bool)
    for _ in range(200):
        p = true_t + tt[..., None] * dirs
        dist = np.minimum.reduce([np.linalg.norm(p - c, axis=-1) - r for c, r in
                                  (((0.3, 0.2, 0.1), 0.5), ((-0.5, 0.4, -0.2), 0.35), ((0.1, -0.6, 0.3), 0.3))])
        hit |= dist < 1e-6
        tt = np.where(hit | (tt > 10), tt, tt + dist / np.linalg.norm(dirs, axis=-1))
    xyz = (tt[..., None] * dirs).astype(np.float32)
    xyz[~hit] = np.nan
    cloud = orc.Cloud(xyz)
    t = orc.CameraTracking(s, 20, 0.001, 1.0, 0.01)
    t.set_K(K)
    t.set_camera_transformation(true_R, true_t + np.array([0.02, -0.015, 0.01]))
    # one Gauss-Newton step removes most of a 2.7 cm offset (voxel = 5 cm; the inverse-L1 field is
    # not smooth, so later steps only wander around the optimum)
    A, b, acc = t.accumulate(s, cloud)
    stop, tw = t.gn_update(A, b)
    assert acc["n_ok"] > 150 and np.linalg.norm(t.trans - true_t) < 0.005
    t.set_camera_transformation(true_R, true_t + np.array([0.02, -0.015, 0.01]))
    st = t.estimate_new_position(s, cloud)
    assert not st["nonfinite"] and 1 <= st["iterations"] <= 20
    assert np.linalg.norm(t.trans - true_t) < 0.02
