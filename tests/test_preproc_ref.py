"""Properties of the NumPy statement of the depth pre-processing (tests/preproc_ref.py), the parity target of the GPU
stage: it has to be a sane filter before bit-equality with it means anything.  CPU only."""
import numpy as np

import preproc_ref as ref
from tracking_sdf_amd import synth

F = np.float32


def noisy_and_clean(w=160, h=120):
    a = synth.Sequence(n_frames=1, width=w, height=h, noise=True, holes=0.03, step=5).frame(0)[0][..., 2].astype(F)
    b = synth.Sequence(n_frames=1, width=w, height=h, noise=False, holes=0.0, step=5).frame(0)[0][..., 2].astype(F)
    return a, b


def test_grid_filter_keeps_a_constant_image_and_the_invalid_mask():
    z = np.full((48, 64), 2.0, dtype=F)
    z[10:14, 20:30] = np.nan
    out = ref.bilateral_grid(z, 15.0, 0.05)
    assert np.array_equal(np.isnan(out), np.isnan(z))
    assert np.array_equal(out[~np.isnan(z)], z[~np.isnan(z)])          # S = 2 W cell by cell (a power of two commutes with every rounding): the quotient is exact
    one = np.full((48, 64), np.nan, dtype=F); one[7, 9] = 0.5
    assert ref.bilateral_grid(one, 4.0, 0.05)[7, 9] == F(0.5)
    z125 = np.full((48, 64), 1.25, dtype=F)
    assert np.allclose(ref.bilateral_grid(z125, 15.0, 0.05), 1.25, rtol=3e-7, atol=0)
    assert np.isnan(ref.bilateral_grid(np.full((8, 8), np.nan, dtype=F), 4.0, 0.05)).all()


def test_grid_filter_is_edge_preserving_and_denoises():
    z = np.full((96, 128), 1.0, dtype=F)
    z[:, 64:] = 2.0                                                      # a 1 m step: 20 sigma_r
    rng = np.random.default_rng(3)
    noisy = (z + rng.normal(0, 0.004, z.shape)).astype(F)
    out = ref.bilateral_grid(noisy, 8.0, 0.05)
    assert np.abs(out - z).max() < 0.012                                 # nothing leaks across the step
    assert np.abs(out - z).mean() < 0.35 * np.abs(noisy - z).mean()      # and the noise is averaged away
    a, b = noisy_and_clean()
    ok = ~np.isnan(a) & ~np.isnan(b)
    g = ref.bilateral_grid(a, 7.5, 0.05)
    wnd = ref.bilateral(a, 15, 7.5, 0.05)
    raw, eg, ew = np.abs(a - b)[ok].mean(), np.abs(g - b)[ok].mean(), np.abs(wnd - b)[ok].mean()
    assert eg < 0.9 * raw and ew < 0.9 * raw                             # both filters beat the raw depth
    assert np.abs(g - wnd)[ok].mean() < 0.003                            # and agree with each other to millimetres


def test_exact_splat_sum_does_not_depend_on_the_pixel_order():
    a, _ = noisy_and_clean(96, 72)
    out = ref.bilateral_grid(a, 6.0, 0.05)
    flipped = ref.bilateral_grid(a[::-1, ::-1].copy(), 6.0, 0.05)[::-1, ::-1]
    # mirrored images have mirrored cells only when (w-1)/sigma_s is an integer; use the splat alone instead:
    # the sums of a cell are integers, so summing the same pixels in reverse gives the same float
    zv = a[~np.isnan(a)]
    q = (zv.astype(np.float64) * 4294967296.0).astype(np.int64)
    assert q.sum() == q[::-1].sum() and np.float32(q.sum() / 4294967296.0) == np.float32(q[::-1].sum() / 4294967296.0)
    assert out.shape == flipped.shape


def test_separable_box_sums_equal_the_window_sums():
    a, _ = noisy_and_clean(96, 72)
    K = synth.default_intrinsics(96, 72)
    xyz = ref.backproject(a, a, K)
    n = ref.normals(xyz, 3, 0.02)
    both = ~np.isnan(n[..., 0])
    assert both.mean() > 0.5
    assert np.allclose(np.linalg.norm(n[both], axis=-1), 1.0, atol=1e-5)
    assert np.all(np.sum(n[both] * xyz[both], axis=-1) <= 0)             # facing the camera
    # brute-force window average of the same gradients, in float64: the unit normals agree to float32 rounding
    h, w, _ = xyz.shape
    r = 3
    dh = np.zeros((h, w, 3)); dv = np.zeros((h, w, 3)); ok = np.zeros((h, w), bool)
    c, l, rr, up, dn = xyz[1:-1, 1:-1], xyz[1:-1, :-2], xyz[1:-1, 2:], xyz[:-2, 1:-1], xyz[2:, 1:-1]
    with np.errstate(invalid="ignore"):
        valid = ~(np.isnan(c[..., 2]) | np.isnan(l[..., 2]) | np.isnan(rr[..., 2]) | np.isnan(up[..., 2]) | np.isnan(dn[..., 2]))
        lim = (F(2.0) * F(0.02) * c[..., 2]).astype(F)
        valid &= (np.abs(rr[..., 2] - l[..., 2]) <= lim) & (np.abs(dn[..., 2] - up[..., 2]) <= lim)
    dh[1:-1, 1:-1] = np.where(valid[..., None], (rr - l).astype(F), 0)
    dv[1:-1, 1:-1] = np.where(valid[..., None], (dn - up).astype(F), 0)
    ok[1:-1, 1:-1] = valid
    ys, xs = np.nonzero(both)
    pick = np.random.default_rng(1).choice(len(ys), 300, replace=False)
    for y, x in zip(ys[pick], xs[pick]):
        y0, y1, x0, x1 = max(0, y - r), min(h, y + r + 1), max(0, x - r), min(w, x + r + 1)
        sh, sv = dh[y0:y1, x0:x1].sum((0, 1)), dv[y0:y1, x0:x1].sum((0, 1))
        m = np.cross(sv, sh)
        m /= np.linalg.norm(m)
        if m @ xyz[y, x] > 0:
            m = -m
        assert np.dot(m, n[y, x]) > 1 - 1e-5


def test_infinite_and_far_depths_are_no_readings():
    """+inf / -inf / absurd ranges are 'no reading' like 0 and NaN; an outlier that needs more depth cells than the grid has
    is dropped from the grid filter and the rest of the image is filtered as if it were not there."""
    rng = np.random.default_rng(3)
    z = (1.5 + 0.01 * rng.standard_normal((40, 50))).astype(np.float32)
    z0 = ref.depth_to_z(z.copy(), 1.0)
    base = ref.bilateral_grid(z0, 4.0, 0.05)
    z[2, 3] = np.inf; z[5, 6] = -np.inf; z[7, 8] = 2000.0; z[9, 10] = 0.0
    zz = ref.depth_to_z(z, 1.0)
    assert np.isnan(zz[2, 3]) and np.isnan(zz[5, 6]) and np.isnan(zz[7, 8]) and np.isnan(zz[9, 10])
    assert np.isfinite(zz).sum() == z.size - 4
    z[20, 30] = 300.0                                   # (300 - 1.5) / 0.05 = 5970 cells > 4091
    zz = ref.depth_to_z(z, 1.0)
    out = ref.bilateral_grid(zz, 4.0, 0.05)
    assert np.isnan(out[20, 30]) and np.isfinite(out).sum() == z.size - 5
    ok = np.isfinite(out)
    assert np.max(np.abs(out[ok] - base[ok])) < 2e-3     # the rest of the image is filtered as before (blur reach aside)
