"""Edge cases of the hot path on the GPU: empty / ragged / tiny inputs, odd sizes, re-sized frames."""
import numpy as np
import pytest

import oracle as orc
from tracking_sdf_amd import synth
from util import make_gpu, make_oracle, sym_rel_err, ulp_diff

pytestmark = pytest.mark.gpu


def render(w, h, k=0, **kw):
    seq = synth.Sequence(n_frames=k + 1, width=w, height=h, noise=True, holes=0.02, step=4, **kw)
    return seq, seq.frame(k)


def compare_update_and_accumulate(m, w, h, vol=None):
    seq, (xyz, nrm, rgb) = render(w, h)
    kw = {} if vol is None else {"vol": vol}
    oo, ot = make_oracle(m, seq.K, **kw)
    go, gt = make_gpu(m, seq.K, **kw)
    n_o = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    st = go.update(gt, xyz, nrm, rgb)
    assert st["n_updated"] == n_o
    D, W = go.download()
    uW = ulp_diff(W, oo.W)
    assert uW.max() <= 1 and ulp_diff(D, oo.D)[uW == 0].max() == 0
    go.upload(oo.D, oo.W)
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
    go.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    assert st_g["n_samples"] == st_o["n_samples"] == ((w + 2) // 3) * ((h + 2) // 3)
    assert st_g["n_terms"] == st_o["n_terms"] and st_g["n_ok"] == st_o["n_ok"]
    if st_o["n_terms"]:
        assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11
    return st_o


@pytest.mark.parametrize("m,w,h", [(33, 161, 121), (50, 100, 75), (64, 17, 13), (40, 64, 2), (31, 3, 200)])
def test_odd_sizes_match_oracle(m, w, h):
    compare_update_and_accumulate(m, w, h)


@pytest.mark.parametrize("w,h,n_wg", [(24, 18, 1), (36, 36, 3), (63, 48, 7), (72, 48, 8), (75, 48, 9), (120, 96, 27)])
def test_tracker_fan_in_with_few_workgroups(w, h, n_wg):
    """The in-launch fan-in of the tracker rows shards the workgroups 8 ways: fewer workgroups than shards, exactly
    8, one more, and an uneven split must all give the oracle's sums -- and the same bits on every repetition,
    whichever workgroup happens to arrive last."""
    assert -(-(((w + 2) // 3) * ((h + 2) // 3)) // 48) == n_wg          # 48 samples per workgroup
    compare_update_and_accumulate(48, w, h)
    seq, (xyz, nrm, rgb) = render(w, h)
    go, gt = make_gpu(48, seq.K)
    go.update(gt, xyz, nrm, rgb)
    A0, b0, st0 = gt.accumulate()
    for _ in range(25):
        A, b, st = gt.accumulate()
        assert np.array_equal(A, A0) and np.array_equal(b, b0) and st == st0


def test_single_pixel_and_2x2_images():
    for w, h in ((1, 1), (2, 2), (4, 1)):
        compare_update_and_accumulate(32, w, h)


def test_all_nan_frame_updates_nothing_and_tracking_reports_no_samples():
    import tracking_sdf_amd as ts
    m = 32
    seq, (xyz, nrm, rgb) = render(64, 48)
    go, gt = make_gpu(m, seq.K)
    go.update(gt, xyz, nrm, rgb)
    D0, W0 = go.download()
    nan = np.full_like(xyz, np.nan)
    st = go.update(gt, nan, nrm, rgb)
    assert st["n_updated"] == 0
    D1, W1 = go.download()
    assert np.array_equal(D0, D1) and np.array_equal(W0, W1)
    pose = (gt.rot.copy(), gt.trans.copy())
    with pytest.raises(ts.TsdfError) as ei:
        gt.estimate_new_position(go, nan)
    assert ei.value.code == ts.E_NO_SAMPLES
    assert np.array_equal(gt.rot, pose[0]) and np.array_equal(gt.trans, pose[1])
    # NaN normals only: still nothing integrated (sdf.cpp:260), tracking unaffected (it never reads normals)
    st = go.update(gt, xyz, np.full_like(nrm, np.nan), rgb)
    assert st["n_updated"] == 0


def test_camera_outside_and_behind_the_volume():
    """Every voxel behind the camera / every sample out of grid: nothing happens, nothing crashes."""
    import tracking_sdf_amd as ts
    m = 32
    seq, (xyz, nrm, rgb) = render(64, 48)
    go, gt = make_gpu(m, seq.K)
    oo, ot = make_oracle(m, seq.K)
    far = np.array([100.0, 100.0, 100.0])
    for t_ in (gt, ot):
        t_.set_camera_transformation(np.eye(3), far)
    assert oo.update(ot, orc.Cloud(xyz, nrm, rgb)) == 0
    assert go.update(gt, xyz, nrm, rgb)["n_updated"] == 0
    go.set_frame(xyz)
    A, b, st = gt.accumulate()
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz))
    assert st["n_oog"] == st_o["n_oog"] > 0 and st["n_terms"] == 0 and not A.any()


def test_frame_size_can_change_between_calls():
    m = 48
    go = oo = None
    seq, f1 = render(96, 72)
    go, gt = make_gpu(m, seq.K)
    oo, ot = make_oracle(m, seq.K)
    for (w, h) in ((96, 72), (160, 120), (40, 30), (160, 120)):
        s2, (xyz, nrm, rgb) = render(w, h)
        gt.set_K(s2.K); ot.set_K(s2.K)
        assert go.update(gt, xyz, nrm, rgb)["n_updated"] == oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    D, W = go.download()
    assert ulp_diff(W, oo.W).max() <= 1


def test_non_cubic_extent_and_shifted_origin():
    vol = dict(width=2.5, height=4.0, depth=1.7, origin=(-1.1, -3.3, 0.1), delta=0.2, epsilon=0.01)
    st = compare_update_and_accumulate(40, 120, 90, vol)
    assert st["n_oog"] > 0            # part of the scene lies outside this volume: the stale carry is active


def test_large_delta_uses_library_exp():
    """delta - epsilon > 0.35 m switches the weight from the Taylor path to exp(): still matches the oracle."""
    vol = dict(width=6.0, height=6.0, depth=3.5, origin=(-3.0, -3.0, -0.5), delta=1.2, epsilon=0.05)
    compare_update_and_accumulate(48, 120, 90, vol)


def test_many_handles_are_created_and_released_cleanly():
    """Create / use / destroy in a loop: device memory comes back (no leak in the handle's many lazily grown buffers)."""
    import torch
    seq, (xyz, nrm, rgb) = render(96, 72)
    free0 = None
    for it in range(12):
        go, gt = make_gpu(40 + (it % 3) * 8, seq.K)
        go.update(gt, xyz, nrm, rgb)
        go.set_frame(xyz)
        gt.accumulate()
        go.mesh(with_color=True)
        go.set_depth_frame((np.nan_to_num(xyz[..., 2]) * 5000).astype(np.uint16), rgb, radius=3)
        go.update()
        go.close()
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        if it == 2:
            free0 = free
        if it > 2:
            assert free >= free0 - (64 << 20), (it, free0, free)      # allocator granularity, not growth


def test_page_locked_caller_buffers_are_copied_from_directly_with_the_same_result():
    """tsdf_set_frame takes page-locked host buffers without the staging memcpy; the buffers are still only borrowed
    for the call (overwriting them right after it must not change the frame)."""
    import torch
    seq, (xyz, nrm, rgb) = render(160, 120)
    a, ta = make_gpu(48, seq.K)
    b, tb = make_gpu(48, seq.K)
    a.update(ta, xyz, nrm, rgb)
    pin = [torch.from_numpy(v).pin_memory() for v in (xyz, nrm, rgb)]
    px, pn_, pc = (t.numpy() for t in pin)
    b.set_frame(px, pn_, pc)
    px[:] = np.nan; pn_[:] = 0; pc[:] = 0                      # the call has returned: the library must not read them any more
    b.update(tb)
    assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a.download(), b.download()))
    assert all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a.download_color(), b.download_color()))
