"""Committed golden vectors (tests/golden/hotpath_m24.npz, made by tools/make_golden.py).

CPU: the oracle must still reproduce them (guards the checker itself).
GPU: the HIP path, through the C ABI, must reproduce them -- no oracle involved at run time.
"""
import os

import numpy as np
import pytest

from util import sym_rel_err, ulp_diff

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_m24.npz"))
VOL = dict(width=2.0, height=3.4, depth=2.0, origin=(-1.0, -3.0, 0.0), delta=0.3, epsilon=0.025)
STAT_KEYS = ("n_samples", "n_nan", "n_oog", "n_fail", "n_ok", "n_terms")


def test_oracle_reproduces_golden():
    import oracle as orc
    m = int(G["m"])
    s = orc.SDF(m, VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])
    t = orc.CameraTracking(s)
    t.set_K(G["K"])
    for k in range(2):
        t.set_camera_transformation(G["R"][k], G["t"][k])
        n = s.update(t, orc.Cloud(G[f"xyz{k}"], G[f"nrm{k}"], G[f"rgb{k}"]), threads=2)
        assert n == G["n_updated"][k]
    for name in ("D", "W", "Color_W", "R", "G", "B"):
        assert np.array_equal(getattr(s, name), G["vol_" + name], equal_nan=True), name
    for p, v, ok in zip(G["probe_pts"], G["probe_val"], G["probe_ok"]):
        got, gok = s.interpolate_distance(p)
        assert gok == bool(ok) and (np.float32(got) == v or (np.isnan(got) and np.isnan(v)))
    cloud = orc.Cloud(G["xyz2"])
    t.set_camera_transformation(G["R"][1], G["t"][1])
    for flag in (1, 0):
        A, b, st = t.accumulate(s, cloud, threads=1, stale_carry=bool(flag))
        assert np.array_equal(A, G[f"A_stale{flag}"]) and np.array_equal(b, G[f"b_stale{flag}"])
        assert [st[k] for k in STAT_KEYS] == list(G[f"acc_stats_stale{flag}"])
    st = t.estimate_new_position(s, cloud, threads=1, stale_carry=True)
    assert st["iterations"] == int(G["track_iterations"])
    assert np.array_equal(t.rot, G["track_rot"]) and np.array_equal(t.trans, G["track_trans"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    import tracking_sdf_amd as ts
    m = int(G["m"])
    for stale in (1, 0):
        s = ts.SDF(m, VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"],
                   stale_carry=bool(stale))
        t = ts.CameraTracking(sdf=s)
        t.set_K(G["K"])
        for k in range(2):
            t.set_camera_transformation(G["R"][k], G["t"][k])
            st = s.update(t, G[f"xyz{k}"], G[f"nrm{k}"], G[f"rgb{k}"])
            assert st["n_updated"] == G["n_updated"][k]
        D, W = s.download()
        cw, r, g, b = s.download_color()
        uW = ulp_diff(W, G["vol_W"])
        assert uW.max() <= 1 and float((uW > 0).mean()) < 1e-3          # exp() last-bit cases only
        for got, name in ((D, "D"), (cw, "Color_W"), (r, "R"), (g, "G"), (b, "B")):
            u = ulp_diff(got, G["vol_" + name])
            assert u[uW == 0].max() == 0 and u.max() <= 4, name
        val, ok = s.interpolate_distance(G["probe_pts"])
        assert np.array_equal(ok, G["probe_ok"])
        same = uW.max() == 0
        if same:
            assert np.array_equal(val, G["probe_val"], equal_nan=True)
        s.upload(G["vol_D"], G["vol_W"])                                # exact golden volume for the tracker part
        t.set_camera_transformation(G["R"][1], G["t"][1])
        s.set_frame(G["xyz2"])
        A, b, st = t.accumulate()
        want = dict(zip(STAT_KEYS, G[f"acc_stats_stale{stale}"]))
        assert st["n_samples"] == want["n_samples"] and st["n_nan"] == want["n_nan"] and st["n_oog"] == want["n_oog"]
        assert st["n_ok"] == want["n_ok"] and st["n_terms"] == want["n_terms"]
        assert st["n_in_grid_owned"] == want["n_ok"] + want["n_fail"]
        assert sym_rel_err(A, G[f"A_stale{stale}"]) < 1e-11 and sym_rel_err(b, G[f"b_stale{stale}"]) < 1e-11
        if stale:
            tr = t.estimate_new_position(s, G["xyz2"])
            assert tr["iterations"] == int(G["track_iterations"]) and tr["stopped"] == int(G["track_stopped"])
            assert np.max(np.abs(t.rot - G["track_rot"])) < 1e-9 and np.max(np.abs(t.trans - G["track_trans"])) < 1e-9
        s.close()


# ---- the visualiser's mesh of the golden volume (tests/golden/mesh_m24.npz, tools/make_golden.py mesh) ----------
GM = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh_m24.npz"))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)


def test_oracle_reproduces_golden_mesh():
    import oracle as orc
    s = orc.SDF(int(G["m"]), VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])
    for name in ("D", "W", "Color_W", "R", "G", "B"):
        getattr(s, name)[:] = G["vol_" + name]
    v, c = s.mesh(with_color=True)
    assert len(v) == len(GM["vertices"]) > 100
    assert np.array_equal(_bits(v), _bits(GM["vertices"])) and np.array_equal(_bits(c), _bits(GM["colors"]))
    assert np.array_equal(_bits(s.mesh(iso_level=0.25)), _bits(GM["vertices_iso025"]))


@pytest.mark.gpu
def test_hip_mesh_reproduces_golden():
    import tracking_sdf_amd as ts
    s = ts.SDF(int(G["m"]), VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])
    s.upload(G["vol_D"], G["vol_W"])
    s.upload_color(G["vol_Color_W"], G["vol_R"], G["vol_G"], G["vol_B"])
    v, c = s.mesh(with_color=True)
    assert np.array_equal(_bits(v), _bits(GM["vertices"])) and np.array_equal(_bits(c), _bits(GM["colors"]))
    assert np.array_equal(_bits(s.mesh(iso_level=0.25)), _bits(GM["vertices_iso025"]))
