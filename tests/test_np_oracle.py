"""The two CPU restatements of the reference's hot path against each other: oracle/tsdf_oracle.c (C, per-voxel loops)
and oracle/np_oracle.py (NumPy, whole-array operations, written from the reference's lines independently of the C
file).  Neither is pinned to the real reference binary (it cannot be built here and ships no vectors) -- this is the
second opinion SURVEY.md section 7 step 1 asks for: a slip of reading or coding in one of them shows up here.

Bars: D, W, colour, interpolation values, A, b (same summation order): bit-exact.  The 6x6 solve uses LAPACK in the
NumPy statement and a hand-written partial-pivot LU in the C one: twist and pose <= 1e-11 relative.
"""
import os

import numpy as np
import pytest

import oracle as orc
from oracle import np_oracle as npo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "hotpath_m24.npz"))
VOL = dict(width=2.0, height=3.4, depth=2.0, origin=(-1.0, -3.0, 0.0), delta=0.3, epsilon=0.025)


def fused_numpy_volume():
    vol = npo.Volume(int(G["m"]), VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])
    trk = npo.Tracker(vol)
    trk.K = np.array(G["K"], dtype=np.float64)
    n_upd = []
    for k in range(2):
        trk.set_camera_transformation(G["R"][k], G["t"][k])
        n_upd.append(npo.update(vol, trk, G[f"xyz{k}"], G[f"nrm{k}"], G[f"rgb{k}"]))
    return vol, trk, n_upd


def test_numpy_update_reproduces_the_golden_volume_bit_for_bit():
    vol, trk, n_upd = fused_numpy_volume()
    assert n_upd == list(G["n_updated"])
    for name in ("D", "W", "Color_W", "R", "G", "B"):
        got, want = getattr(vol, name), G["vol_" + name]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    assert (vol.W > 0).sum() > 1000 and ((vol.W > 0) & (vol.W != np.round(vol.W))).sum() > 20      # the exp() band is exercised


def test_numpy_interpolation_matches_golden_probes():
    vol, _, _ = fused_numpy_volume()
    val, ok = vol.interpolate_distance(G["probe_pts"])
    assert np.array_equal(ok, G["probe_ok"])
    assert np.array_equal(val[ok].view(np.uint32), G["probe_val"][ok].view(np.uint32))
    assert np.all(np.isnan(val[~ok]))
    assert ok.sum() > 50 and (~ok).sum() > 50


def test_numpy_interpolation_kats():
    """SURVEY 8c KAT-2, derived by hand from sdf.cpp:127-163: inverse-L1 weights, exact-hit return, (int) truncation."""
    m = 8
    vol = npo.Volume(m, 1.0, 1.0, 1.0, (0, 0, 0), 0.3, 0.025)
    vol.W[:] = 1
    i, j, k = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    vol.D[:] = j.reshape(-1).astype(np.float32)
    v, ok = vol.interpolate_distance([[3.5, 2.0, 4.0], [3.0, 2.0, 4.0], [-0.5, 2.0, 4.0], [20.0, 2.0, 4.0]])
    assert ok.tolist() == [True, True, True, False]
    assert abs(float(v[0]) - (2.0 + 2.0 / 7.0)) < 1e-6         # not trilinear (that would be 2)
    assert v[1] == 2.0                                         # exact hit returns the corner
    assert np.isfinite(v[2]) and np.isnan(v[3])                # -0.5 truncates to base 0; no valid corner -> 0/0
    vol.W[:] = 0
    idx = (m * m) * 3 + m * 2 + 4
    vol.W[idx] = 2
    vol.D[idx] = 7.25
    v, ok = vol.interpolate_distance([[3.4, 2.3, 4.6]])
    assert ok[0] and abs(float(v[0]) - 7.25) < 1e-6            # one valid corner: (w D) / w, its value up to rounding


@pytest.mark.parametrize("stale", [1, 0])
def test_numpy_accumulate_matches_golden_normal_equations(stale):
    vol, trk, _ = fused_numpy_volume()
    trk.set_camera_transformation(G["R"][1], G["t"][1])
    A, b, st = npo.accumulate(vol, trk, G["xyz2"], stale_carry=bool(stale))
    want = G[f"acc_stats_stale{stale}"]
    assert [st[k] for k in ("n_samples", "n_nan", "n_oog", "n_fail", "n_ok", "n_terms")] == want.tolist()
    assert np.array_equal(A, G[f"A_stale{stale}"]) and np.array_equal(b, G[f"b_stale{stale}"])
    if stale:
        assert st["n_terms"] > st["n_ok"]                      # the carry-over fires in this fixture


def test_numpy_tracker_reaches_the_golden_pose():
    vol, trk, _ = fused_numpy_volume()
    trk.set_camera_transformation(G["R"][1], G["t"][1])
    st = npo.estimate_new_position(vol, trk, G["xyz2"], stale_carry=True)
    assert st["iterations"] == int(G["track_iterations"]) and st["stopped"] == int(G["track_stopped"])
    scale = max(1.0, float(np.max(np.abs(G["track_twist"]))))
    assert np.max(np.abs(trk.rot - G["track_rot"])) < 1e-11 and np.max(np.abs(trk.trans - G["track_trans"])) < 1e-11
    assert np.max(np.abs(st["last_twist"] - G["track_twist"])) < 1e-11 * scale


def test_numpy_exp_map_and_pose_algebra_match_the_c_oracle():
    rng = np.random.default_rng(5)
    L = orc.lib()
    for tw in [np.zeros(6), np.array([1.0, 0, 0, 0, 0, np.pi / 2]), np.array([0.1, -0.2, 0.3, 1e-5, -2e-5, 1e-5])] + \
            [rng.normal(size=6) * s for s in (1e-3, 1e-1, 1.0)]:
        R, t = npo.exp_map(tw, 1.0)
        out = np.zeros(12)
        L.orc_direct_exponential_map(tw.ctypes.data_as(orc.C.POINTER(orc.C.c_double)), 1.0,
                                     out.ctypes.data_as(orc.C.POINTER(orc.C.c_double)))
        Rc, tc = out.reshape(3, 4)[:, :3], out.reshape(3, 4)[:, 3]
        assert np.array_equal(R, Rc) and np.array_equal(t, tc)
    R, t = npo.exp_map([1.0, 0, 0, 0, 0, np.pi / 2])            # KAT-4
    assert np.allclose(R, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15) and np.allclose(t, [2 / np.pi, 2 / np.pi, 0])
    M = rng.normal(size=(3, 3))
    out = np.zeros(9)
    L.orc_inverse3(np.ascontiguousarray(M).ctypes.data_as(orc.C.POINTER(orc.C.c_double)),
                   out.ctypes.data_as(orc.C.POINTER(orc.C.c_double)))
    assert np.array_equal(npo.inverse3(M), out.reshape(3, 3))


def test_numpy_perturbed_rotations_and_steps_match_the_c_oracle():
    so = orc.SDF(24, VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])
    to = orc.CameraTracking(so, 20, 0.001, 1.0, 0.01)
    vol = npo.Volume(24, VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])
    trk = npo.Tracker(vol, 20, 0.001, 1.0, 0.01)                 # KAT-8: definition order (max_iter, max_twist, v_h, w_h)
    assert trk.v_h == np.float32(1.0) and trk.w_h == np.float32(0.01)
    to.set_camera_transformation(G["R"][1], G["t"][1])
    trk.set_camera_transformation(G["R"][1], G["t"][1])
    assert np.array_equal(np.array(trk.perturbed_rotations()), to.perturbed_rotations())
    assert np.array_equal(trk.rot_inv, to.rot_inv) and np.array_equal(trk.rot_inv_trans, to.rot_inv_trans)
    t_ = to._p.contents
    assert (trk.v_h2_width, trk.v_h2_height, trk.v_h2_depth) == (t_.v_h2_width, t_.v_h2_height, t_.v_h2_depth)
