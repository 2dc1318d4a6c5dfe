"""CPU tests of the synthetic RGB-D stream (test / benchmark infrastructure): the two renderers agree, the scene
keeps the camera path clear, and frames have the conventions the hot path expects."""
import numpy as np

from tracking_sdf_amd import synth


def test_torch_renderer_matches_numpy_renderer():
    seq = synth.Sequence(n_frames=1246, width=96, height=72)
    for k in (0, 137, 333, 700, 951, 1245):
        x, n, c = seq.frame(k)
        xt, nt, ct = (a.numpy() for a in seq.frame_torch(k, "cpu"))
        assert x.dtype == xt.dtype == np.float32 and c.dtype == ct.dtype == np.uint8
        assert np.array_equal(np.isnan(x), np.isnan(xt))
        assert np.nanmax(np.abs(x - xt)) < 1e-6 and np.nanmax(np.abs(n - nt)) < 1e-6
        assert np.array_equal(c, ct)


def test_torch_renderer_noise_is_reproducible_and_kinect_sized():
    seq = synth.Sequence(n_frames=30, width=96, height=72, noise=True, holes=0.02, seed=3)
    a = [t.numpy().copy() for t in seq.frame_torch(20, "cpu")]
    b = [t.numpy().copy() for t in seq.frame_torch(20, "cpu")]
    assert all(np.array_equal(p, q, equal_nan=True) for p, q in zip(a, b))
    clean = synth.Sequence(n_frames=30, width=96, height=72).frame(20)[0]
    z, z0 = a[0][..., 2], clean[..., 2]
    both = ~np.isnan(z) & ~np.isnan(z0)
    holes = np.isnan(z) & ~np.isnan(z0)
    assert 0.005 < holes.mean() < 0.05
    err = (z - z0)[both]
    assert 0.0005 < err.std() < 0.02 and abs(err.mean()) < 0.002


def test_camera_path_keeps_sensor_distance_to_every_object():
    """The whole ground-truth path stays >= 0.4 m (the sensor's minimum range) away from every primitive, so no
    frame loses a large part of its pixels to the near clip, and inside the room."""
    _, _, t = synth.load_trajectory()
    assert np.all(t > synth.ROOM_LO + 0.4) and np.all(t < synth.ROOM_HI - 0.4)
    for c, r, _ in synth.SPHERES:
        assert (np.linalg.norm(t - np.array(c), axis=1) - r).min() > 0.4
    for lo, hi, _ in synth.BOXES:
        q = np.maximum(np.maximum(np.array(lo) - t, t - np.array(hi)), 0)
        assert np.linalg.norm(q, axis=1).min() > 0.4
    for cx, cy, r, z0, z1, _ in synth.CYLINDERS:
        rad = np.maximum(np.hypot(t[:, 0] - cx, t[:, 1] - cy) - r, 0)
        dz = np.maximum(np.maximum(z0 - t[:, 2], t[:, 2] - z1), 0)
        assert np.hypot(rad, dz).min() > 0.4


def test_the_plant_stands_where_the_camera_looks():
    """fr1/plant circles its subject: most frames must see the plant (near geometry), not only far walls."""
    seq = synth.Sequence(n_frames=1246, width=64, height=48)
    near = []
    for k in range(0, 1246, 40):
        z = seq.frame(k)[0][..., 2]
        assert np.isnan(z).mean() < 0.05
        near.append(np.nanmean(z < 1.2))
    assert np.median(near) > 0.08


def test_normals_face_the_camera_and_are_unit():
    seq = synth.Sequence(n_frames=400, width=64, height=48)
    for k in (0, 200, 399):
        x, n, _ = seq.frame(k)
        ok = ~np.isnan(x[..., 0])
        assert np.allclose(np.linalg.norm(n[ok], axis=-1), 1.0, atol=1e-5)
        assert ((n[ok] * x[ok]).sum(-1) <= 1e-6).all()        # n . (P - 0) <= 0: oriented toward the camera at 0
