"""GPU depth pre-processing (tsdf_set_depth_frame) against its NumPy statement (tests/preproc_ref.py).
Tolerances: x,y bit-exact; z bit-exact with the bilateral grid (grid_filter=1, the default), within 2e-6 relative
with the windowed filter (expf differs in the last bits between the GPU and NumPy); normals within 2e-4 (angle)
where both are defined, identical NaN masks away from ties."""
import numpy as np
import pytest

import preproc_ref as ref
from tracking_sdf_amd import synth

pytestmark = pytest.mark.gpu


def depth_image(w, h, k=0, to_u16=True):
    seq = synth.Sequence(n_frames=k + 1, width=w, height=h, noise=True, holes=0.03, step=5)
    xyz, nrm, rgb = seq.frame(k)
    z = xyz[..., 2]
    if to_u16:
        d = np.where(np.isnan(z), 0, np.round(z * 5000.0)).astype(np.uint16)
    else:
        d = np.where(np.isnan(z), 0.0, z).astype(np.float32)
    return seq, d, rgb


@pytest.mark.parametrize("w,h,params", [
    (96, 72, dict(radius=4, sigma_s=2.0, sigma_r=0.03, normal_radius=2, grid_filter=0)),
    (160, 120, dict(radius=9, sigma_s=4.5, sigma_r=0.05, normal_radius=5, grid_filter=0)),
    (64, 48, dict(grid_filter=0)),                      # windowed, PCL-like sigmas: sigma_s 15, radius 30
    (80, 60, dict(radius=0, normal_radius=1)),          # no filtering
    (64, 48, dict()),                                   # the defaults: bilateral grid, sigma_s 15, sigma_r 0.05
    (160, 120, dict(sigma_s=4.5, sigma_r=0.05)),
    (203, 117, dict(sigma_s=7.3, sigma_r=0.021, normal_radius=3)),   # ragged cells, many depth cells
    (96, 72, dict(sigma_s=1.0, sigma_r=0.5)),           # one pixel per cell
    (320, 240, dict(sigma_s=30.0, sigma_r=0.05)),       # the largest cell (961 pixels)
])
def test_preprocessing_matches_numpy_statement(w, h, params):
    import tracking_sdf_amd as ts
    seq, d16, rgb = depth_image(w, h)
    s = ts.SDF(32)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    s.set_depth_frame(d16, rgb, **params)
    xyz, nrm = s.get_preprocessed()
    want_xyz, want_n = ref.preprocess(d16, seq.K, **params)
    assert np.array_equal(np.isnan(xyz[..., 2]), np.isnan(want_xyz[..., 2]))
    ok = ~np.isnan(want_xyz[..., 2])
    assert np.array_equal(xyz[..., :2][ok], want_xyz[..., :2][ok])                 # raw back-projection: exact
    if params.get("grid_filter", 1) and params.get("radius", 30) > 0:
        assert np.array_equal(xyz[..., 2][ok], want_xyz[..., 2][ok])               # bilateral grid: exact
    assert np.max(np.abs(xyz[..., 2][ok] - want_xyz[..., 2][ok]) / want_xyz[..., 2][ok]) < 2e-6
    both = ~np.isnan(nrm[..., 0]) & ~np.isnan(want_n[..., 0])
    assert (np.isnan(nrm[..., 0]) != np.isnan(want_n[..., 0])).mean() < 2e-3        # discontinuity-test ties only
    assert both.mean() > 0.5
    cosang = np.sum(nrm[both] * want_n[both], axis=-1)
    assert np.min(cosang) > 1.0 - 2e-6 or np.mean(cosang < 1.0 - 2e-8) < 1e-2
    assert np.allclose(np.linalg.norm(nrm[both], axis=-1), 1.0, atol=1e-5)
    assert np.all(np.sum(nrm[both] * xyz[both], axis=-1) <= 0)                      # facing the camera
    s.close()


def test_float_depth_input_and_pipeline_runs_from_depth_only():
    """Depth-only input drives the whole hot path: integrate, then track the next frame."""
    import tracking_sdf_amd as ts
    w, h, m = 160, 120, 64
    seq = synth.Sequence(n_frames=3, width=w, height=h, noise=True, holes=0.02, step=3)
    s = ts.SDF(m)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    params = dict(radius=6, sigma_s=3.0, sigma_r=0.05, normal_radius=3)
    for k in range(3):
        xyz, nrm, rgb = seq.frame(k)
        z = np.where(np.isnan(xyz[..., 2]), 0.0, xyz[..., 2]).astype(np.float32)
        s.set_depth_frame(z, rgb, **params)
        if k > 0:
            st = t.estimate_new_position()
            assert 1 <= st["iterations"] <= 20
        st = s.update()
        assert st["n_updated"] > 1000
    assert np.linalg.norm(t.trans - seq.t[2]) < 0.08           # follows the true path (voxel = 9 cm here)
    # the smoothed normals agree with the analytic ones of the synthetic scene on the big planes
    xyz_p, nrm_p = s.get_preprocessed()
    ana = seq.frame(2)[1]
    both = ~np.isnan(nrm_p[..., 0]) & ~np.isnan(ana[..., 0])
    cosang = np.sum(nrm_p[both] * ana[both], axis=-1)
    assert np.median(cosang) > 0.99
    s.close()


@pytest.mark.parametrize("grid_filter", [1, 0])
def test_infinite_and_far_depths_are_no_readings(grid_filter):
    """ROS float depth images mark 'too far' with +inf (REP-117); one such pixel, or one at 2000 m, must not cost the
    frame: they are treated like NaN / 0.  A finite outlier whose range needs more depth cells than a cell column of
    the bilateral grid holds (here 300 m at sigma_r 0.05) drops out of the filter, the near scene is still filtered --
    all of it exactly as the NumPy statement does."""
    import tracking_sdf_amd as ts
    w, h = 96, 72
    seq, z, rgb = depth_image(w, h, to_u16=False)
    z = z.copy()
    z[3, 5] = np.inf; z[4, 7] = -np.inf; z[10, 11] = 2000.0; z[20, 30] = 300.0; z[21, 31] = np.nan
    s = ts.SDF(32)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    params = dict(sigma_s=4.0, sigma_r=0.05, radius=6, normal_radius=2, grid_filter=grid_filter)
    s.set_depth_frame(z, rgb, **params)
    xyz, nrm = s.get_preprocessed()
    want_xyz, want_n = ref.preprocess(z, seq.K, depth_scale=1.0, **params)
    for (r, c) in ((3, 5), (4, 7), (10, 11), (21, 31)):
        assert np.isnan(xyz[r, c, 2])
    assert np.array_equal(np.isnan(xyz[..., 2]), np.isnan(want_xyz[..., 2]))
    ok = ~np.isnan(want_xyz[..., 2])
    assert ok.mean() > 0.8
    if grid_filter:
        assert np.isnan(xyz[20, 30, 2])                       # beyond the grid's depth cells: dropped
        assert np.array_equal(xyz[..., 2][ok], want_xyz[..., 2][ok])
    else:
        assert np.max(np.abs(xyz[..., 2][ok] - want_xyz[..., 2][ok]) / want_xyz[..., 2][ok]) < 2e-6
    s.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_queued_depth_frames_give_the_same_trajectory_and_volume(pinned):
    """tsdf_queue_depth_frame: frame k+1 is uploaded, pre-processed and packed while frame k is tracked and integrated;
    poses, volume and the pre-processed planes must equal the tsdf_set_depth_frame loop bit for bit; a frame whose
    pre-processing fails on the library thread reports it from tsdf_next_frame."""
    import torch
    import tracking_sdf_amd as ts
    w, h, m, n = 160, 120, 48, 5
    seq = synth.Sequence(n_frames=n, width=w, height=h, noise=True, holes=0.02, step=3)
    params = dict(sigma_s=3.0, sigma_r=0.05, normal_radius=3)
    frames, hold = [], []
    for k in range(n):
        xyz, _, rgb = seq.frame(k)
        z16 = np.clip(np.where(np.isnan(xyz[..., 2]), 0.0, xyz[..., 2]) * 5000.0, 0, 65535).astype(np.uint16)
        if pinned:
            tz, tc = torch.from_numpy(z16.view(np.int16).copy()).pin_memory(), torch.from_numpy(np.ascontiguousarray(rgb)).pin_memory()
            hold.append((tz, tc))
            frames.append((tz.numpy().view(np.uint16), tc.numpy()))
        else:
            frames.append((z16, np.ascontiguousarray(rgb)))

    def run(queued, ahead=1):
        s = ts.SDF(m, with_color=True)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        poses, pres = [], []
        if queued:
            for j in range(ahead):
                s.queue_depth_frame(*frames[j], depth_scale=1.0 / 5000.0, **params)
        for k in range(n):
            if queued:
                s.next_frame()
                if k + ahead < n:
                    s.queue_depth_frame(*frames[k + ahead], depth_scale=1.0 / 5000.0, **params)
                    if ahead == 2:
                        with pytest.raises(ts.TsdfError):
                            s.queue_depth_frame(*frames[k + ahead], depth_scale=1.0 / 5000.0, **params)   # current + 2: full
                    with pytest.raises(ts.TsdfError):
                        s.set_depth_frame(*frames[k], depth_scale=1.0 / 5000.0, **params)      # a frame is queued
            else:
                s.set_depth_frame(*frames[k], depth_scale=1.0 / 5000.0, **params)
            # (with frame k+1 queued behind it: since round 6 a queued frame has a device block of its own and frame k's
            # planes stay readable; until round 5 this call failed with E_NO_FRAME)
            pres.append(s.get_preprocessed())
            if k > 0:
                t.estimate_new_position()
            s.update()
            poses.append((t.rot.copy(), t.trans.copy()))
        pre = s.get_preprocessed()
        out = (poses, s.download(), s.download_color(), pre, pres)
        s.close()
        return out
    want = run(False)
    for got in (run(True), run(True, ahead=2)):             # one frame waiting, two frames waiting
        for (r0, t0), (r1, t1) in zip(want[0], got[0]):
            assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
        for a, b in zip(want[1] + want[2], got[1] + got[2]):
            assert np.array_equal(a, b)
        for a, b in zip(want[3], got[3]):
            assert np.array_equal(a, b, equal_nan=True)
        for pa, pb in zip(want[4], got[4]):                 # every frame's pre-processed planes, read while it was current
            for a, b in zip(pa, pb):
                assert np.array_equal(a, b, equal_nan=True)
    # a depth range that is no usable bilateral grid: refused by the frame's tsdf_next_frame, the current frame stays
    s = ts.SDF(32)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    s.set_depth_frame(*frames[0], depth_scale=1.0 / 5000.0, **params)
    bad = np.linspace(1.0, 5.0, h * w, dtype=np.float32).reshape(h, w)     # 4000 depth cells of 1 mm x 164 x 124: too many
    s.queue_depth_frame(bad, None, sigma_s=1.0, sigma_r=0.001, normal_radius=3)
    with pytest.raises(ts.TsdfError) as ei:
        s.next_frame()
    assert "too large" in str(ei.value)
    s.update()                                       # frame 0 is still the current one
    s.close()


def test_a_bad_frame_behind_a_good_one_in_the_queue_is_reported_by_its_own_next_frame():
    """Two depth frames waiting, the second one with a depth range that is no usable bilateral grid: the first
    tsdf_next_frame succeeds, the second reports the second frame's error (and its message), the current frame stays, and the
    queue goes on working."""
    import tracking_sdf_amd as ts
    w, h = 160, 120
    seq = synth.Sequence(n_frames=3, width=w, height=h, noise=True, holes=0.02, step=3)
    params = dict(sigma_s=3.0, sigma_r=0.05, normal_radius=3)
    z16 = []
    for k in range(3):
        xyz = seq.frame(k)[0]
        z16.append(np.clip(np.where(np.isnan(xyz[..., 2]), 0.0, xyz[..., 2]) * 5000.0, 0, 65535).astype(np.uint16))
    bad = np.linspace(1.0, 5.0, h * w, dtype=np.float32).reshape(h, w)
    s = ts.SDF(32, with_color=False)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    s.queue_depth_frame(z16[0], None, depth_scale=1.0 / 5000.0, **params)
    s.queue_depth_frame(bad, None, sigma_s=1.0, sigma_r=0.001, normal_radius=3)
    s.next_frame()
    s.update()
    first = s.get_preprocessed()
    with pytest.raises(ts.TsdfError) as ei:
        s.next_frame()
    assert "too large" in str(ei.value)
    again = s.get_preprocessed()                     # frame 0 is still the current one
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(first, again))
    s.queue_depth_frame(z16[1], None, depth_scale=1.0 / 5000.0, **params)
    s.queue_depth_frame(z16[2], None, depth_scale=1.0 / 5000.0, **params)
    for _ in range(2):
        s.next_frame()
        t.estimate_new_position()
        s.update()
    with pytest.raises(ts.TsdfError):
        s.next_frame()
    s.close()


def test_preproc_argument_checks():
    import tracking_sdf_amd as ts
    s = ts.SDF(32)
    d = np.zeros((48, 64), dtype=np.uint16)
    with pytest.raises(ts.TsdfError) as ei:
        s.set_depth_frame(d)                               # intrinsics missing
    assert ei.value.code == ts.E_NO_INTRINSICS
    t = ts.CameraTracking(sdf=s)
    t.set_K(synth.default_intrinsics(64, 48))
    with pytest.raises(ts.TsdfError):
        s.set_depth_frame(d, radius=40)
    for bad in (dict(sigma_s=0.5), dict(sigma_s=31.0), dict(depth_scale=0.0)):
        with pytest.raises(ts.TsdfError) as ei:
            s.set_depth_frame(d, **bad)
        assert ei.value.code == ts.E_BADARG
    deep = np.zeros((48, 64), dtype=np.float32); deep[0, 0] = 1.0; deep[1, 1] = np.inf
    s.set_depth_frame(deep)                                # +inf is 'no reading' (REP-117), not an infinite depth range
    xyzd, _ = s.get_preprocessed()
    assert xyzd[0, 0, 2] == np.float32(1.0) and np.isnan(xyzd[..., 2]).sum() == 48 * 64 - 1
    one = np.zeros((48, 64), dtype=np.uint16); one[20, 30] = 5000
    s.set_depth_frame(one)                                 # a single valid pixel filters to itself
    xyz1, _ = s.get_preprocessed()
    assert xyz1[20, 30, 2] == np.float32(5000) * np.float32(1.0 / 5000.0) and np.isnan(xyz1[..., 2]).sum() == 48 * 64 - 1
    s.set_depth_frame(d, radius=2)                         # all-invalid depth: fine, everything NaN
    xyz, nrm = s.get_preprocessed()
    assert np.isnan(xyz).all() and np.isnan(nrm).all()
    s.close()
