"""Randomised parity: the HIP path against the oracle over random volume placements, poses, image sizes, slabs and
both stale-carry modes (TSDF_FUZZ_CASES=400 was run clean on MI355X) -- aimed at the corner cases of the paired corner loads (look-ups that straddle the grid
border in k), the row clipping and the stale-carry multiplicities.  Seeded: every run sees the same cases."""
import numpy as np
import pytest

import oracle as orc
import tracking_sdf_amd as ts
from tracking_sdf_amd import synth
from util import make_gpu, make_oracle, sym_rel_err, ulp_diff

pytestmark = pytest.mark.gpu


def random_rotation(rng, max_angle):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    a = rng.uniform(-max_angle, max_angle)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * (Kx @ Kx)


import os


@pytest.mark.parametrize("case", range(int(os.environ.get("TSDF_FUZZ_CASES", "24"))))
def test_random_configuration_matches_oracle(case):
    rng = np.random.default_rng(1000 + case)
    m = int(rng.integers(9, 44))
    w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
    seq = synth.Sequence(n_frames=3, width=w, height=h, noise=bool(case & 1), holes=0.05 * (case % 3), step=int(rng.integers(1, 9)))
    # a volume that covers only part of the scene, so that samples leave the grid and look-ups cross its faces
    ext = rng.uniform(1.0, 5.0, size=3)
    org = np.array([rng.uniform(-3.0, 0.5), rng.uniform(-3.0, -0.5), rng.uniform(-0.5, 1.0)])
    vol = dict(width=float(ext[0]), height=float(ext[1]), depth=float(ext[2]), origin=tuple(org),
               delta=float(rng.uniform(0.05, 0.5)), epsilon=float(rng.uniform(0.005, 0.04)))
    gn = (20, 0.001, float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([0.01, 0.002, 0.03])))
    stale = bool(rng.integers(0, 2))
    oo, ot = make_oracle(m, seq.K, vol=vol, gn=gn)
    go, gt = make_gpu(m, seq.K, vol=vol, gn=gn, stale_carry=stale)
    n_upd = 0
    for k in range(2):
        xyz, nrm, rgb = seq.frame(k)
        R = seq.R[k] @ random_rotation(rng, 0.05)
        t = seq.t[k] + rng.normal(scale=0.02, size=3)
        for trk in (ot, gt):
            trk.set_camera_transformation(R, t)
        n_o = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
        st = go.update(gt, xyz, nrm, rgb)
        assert st["n_updated"] == n_o, (case, k)
        n_upd += n_o
    D, W = go.download()
    uW = ulp_diff(W, oo.W)
    assert uW.max() <= 1
    if (uW == 0).any():
        assert ulp_diff(D, oo.D)[uW == 0].max() == 0
    cw, r, g, b = go.download_color()
    for got, want in ((cw, oo.Color_W), (r, oo.R), (g, oo.G), (b, oo.B)):
        assert ulp_diff(got, want)[uW == 0].max() <= 2
    # tracker pass on the oracle's exact volume, at a perturbed pose, whole volume and as two slabs
    go.upload(oo.D, oo.W)
    xyz2 = seq.frame(2)[0]
    R = seq.R[1] @ random_rotation(rng, 0.03)
    t = seq.t[1] + rng.normal(scale=0.03, size=3)
    for trk in (ot, gt):
        trk.set_camera_transformation(R, t)
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz2), threads=1, stale_carry=stale)
    go.set_frame(xyz2)
    A_g, b_g, st_g = gt.accumulate()
    for key in ("n_samples", "n_nan", "n_oog", "n_ok", "n_terms"):
        assert st_g[key] == st_o[key], (case, key)
    if st_o["n_terms"]:
        assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11
    # interpolation probes around and across the grid faces
    pts = rng.uniform(-2.0, m + 1.0, size=(256, 3))
    val, ok = go.interpolate_distance(pts)
    for p_, v_, k_ in zip(pts, val, ok):
        vo, ko = oo.interpolate_distance(p_)
        assert bool(k_) == ko and (np.float32(vo) == v_ or (np.isnan(vo) and np.isnan(v_)))
    # two x-slabs with a generous halo: partial sums add up to the whole, counters too
    if m >= 16:
        cut = int(rng.integers(4, m - 4))
        halo = ts.halo_for(go.cfg, 8.0)
        A_s, b_s, terms = np.zeros((6, 6)), np.zeros(6), 0
        for x0, x1 in ((0, cut), (cut, m)):
            gs, gts = make_gpu(m, seq.K, vol=vol, gn=gn, stale_carry=stale, slab=(x0, x1), halo=halo)
            gs.upload_with_halo(oo.D, oo.W)
            gts.set_camera_transformation(R, t)
            gs.set_frame(xyz2)
            A_p, b_p, st_p = gts.accumulate()
            A_s += A_p; b_s += b_p; terms += st_p["n_terms"]
            gs.close()
        assert terms == st_o["n_terms"]
        if terms:
            assert sym_rel_err(A_s, A_o) < 1e-11 and sym_rel_err(b_s, b_o) < 1e-11
    go.close()
