"""NumPy statement of this repository's depth pre-processing stand-in (the parity target of
tracking_sdf_amd/csrc/preproc_kernels.hip).  TEST INFRASTRUCTURE ONLY.

The reference pre-processes with PCL (sdf_reconstruction.cpp:29-49), which is neither in the reference tree
nor in this image: PARITY WITH PCL IS UNPINNED.  Every float operation below is done in float32, in the same
order as the kernels, so the two agree to the last bits except for expf; the bilateral-grid
filter (grid_filter=1, the default) has no transcendental in it and agrees bit for bit.
"""
import numpy as np

F = np.float32


MAX_DEPTH = F(1000.0)          # a depth is a measurement when 0 < z < MAX_DEPTH (zero / negative / NaN / +inf / absurd: no reading)
SPLAT_MAX_DEPTH_CELLS = 4096


def depth_to_z(depth, scale):
    if depth.dtype == np.uint16:
        z = depth.astype(F) * F(scale)
    else:
        z = depth.astype(F).copy()
    with np.errstate(invalid="ignore"):
        z[~((z > 0) & (z < MAX_DEPTH))] = np.nan
    return z


def bilateral(z, R, sigma_s, sigma_r):
    if R == 0:
        return z.copy()
    h, w = z.shape
    inv2ss = F(1.0) / (F(2.0) * F(sigma_s) * F(sigma_s))
    inv2sr = F(1.0) / (F(2.0) * F(sigma_r) * F(sigma_r))
    pad = np.full((h + 2 * R, w + 2 * R), np.nan, dtype=F)
    pad[R:R + h, R:R + w] = z
    num = np.zeros((h, w), dtype=F)
    den = np.zeros((h, w), dtype=F)
    with np.errstate(invalid="ignore"):
        for dy in range(-R, R + 1):
            for dx in range(-R, R + 1):
                zq = pad[R + dy:R + dy + h, R + dx:R + dx + w]
                dz = (zq - z).astype(F)
                arg = (-(F(dx * dx + dy * dy) * inv2ss) - (dz * dz).astype(F) * inv2sr).astype(F)
                wgt = np.exp(arg, dtype=F)
                ok = ~np.isnan(zq)
                num = np.where(ok, (num + (wgt * zq).astype(F)).astype(F), num)
                den = np.where(ok, (den + wgt).astype(F), den)
        out = (num / den).astype(F)
    out[np.isnan(z)] = np.nan
    return out


GRID_PAD = 2


def _blur_axis(G, axis):
    """Two passes of [1 2 1]/4 along `axis` over the cells interior in all three axes; face cells stay."""
    for _ in range(2):
        N = G.copy()
        inner = G[1:-1, 1:-1, 1:-1]
        lo = [slice(1, -1)] * 3; hi = [slice(1, -1)] * 3
        lo[axis] = slice(0, -2); hi[axis] = slice(2, None)
        N[1:-1, 1:-1, 1:-1] = (((G[tuple(lo)] + G[tuple(hi)]).astype(F) + (F(2.0) * inner).astype(F)).astype(F) * F(0.25)).astype(F)
        G = N
    return G


def bilateral_grid(z, sigma_s, sigma_r):
    """Bilateral grid (Paris & Durand 2006; the algorithm family of pcl::FastBilateralFilter): nearest-cell splat of
    (z, 1) with exact depth sums, [1 2 1]/4 twice per axis, trilinear slice, z' = S / W."""
    h, w = z.shape
    P = GRID_PAD
    ss, sr = F(sigma_s), F(sigma_r)
    valid = ~np.isnan(z)
    if not valid.any():
        return z.copy()
    zv = z[valid]
    zmin, zmax = zv.min(), zv.max()
    nz_limit = F(SPLAT_MAX_DEPTH_CELLS - 1 - 2 * P)
    if not F(zmax - zmin) / sr < nz_limit:
        # the range does not fit the depth cells of a cell column: keep the near part, drop the pixels behind it
        zmax = F(zmin + F(sr * F(nz_limit - F(1.0))))
        with np.errstate(invalid="ignore"):
            valid = valid & (z < zmax)
        zv = z[valid]
    gx = int(F(w - 1) / ss) + 1 + 2 * P
    gy = int(F(h - 1) / ss) + 1 + 2 * P
    gz = int(F(zmax - zmin) / sr) + 1 + 2 * P
    ys, xs = np.nonzero(valid)
    cx = (xs.astype(F) / ss + F(0.5)).astype(np.int64) + P
    cy = (ys.astype(F) / ss + F(0.5)).astype(np.int64) + P
    cz = ((zv - zmin).astype(F) / sr + F(0.5)).astype(np.int64) + P
    Si = np.zeros((gx, gy, gz), dtype=np.int64)                 # exact sums of trunc(z * 2^32), rounded to f32 once
    Wi = np.zeros((gx, gy, gz), dtype=np.int64)
    np.add.at(Si, (cx, cy, cz), (zv.astype(np.float64) * 4294967296.0).astype(np.int64))
    np.add.at(Wi, (cx, cy, cz), 1)
    S = (Si.astype(np.float64) * (1.0 / 4294967296.0)).astype(F)
    W = Wi.astype(F)
    for axis in range(3):
        S = _blur_axis(S, axis)
        W = _blur_axis(W, axis)
    px = (xs.astype(F) / ss + F(P)).astype(F)
    py = (ys.astype(F) / ss + F(P)).astype(F)
    pz = ((zv - zmin).astype(F) / sr + F(P)).astype(F)
    xi, yi, zi = px.astype(np.int64), py.astype(np.int64), pz.astype(np.int64)
    xa, ya, za = (px - xi.astype(F)).astype(F), (py - yi.astype(F)).astype(F), (pz - zi.astype(F)).astype(F)
    xi, yi, zi = np.clip(xi, 0, gx - 1), np.clip(yi, 0, gy - 1), np.clip(zi, 0, gz - 1)
    xx, yy, zz = np.minimum(xi + 1, gx - 1), np.minimum(yi + 1, gy - 1), np.minimum(zi + 1, gz - 1)
    xb, yb, zb = (F(1.0) - xa).astype(F), (F(1.0) - ya).astype(F), (F(1.0) - za).astype(F)
    accS = accW = None
    for (wx, ix), (wy, iy), (wz, iz) in [((xb, xi), (yb, yi), (zb, zi)), ((xa, xx), (yb, yi), (zb, zi)),
                                          ((xb, xi), (ya, yy), (zb, zi)), ((xa, xx), (ya, yy), (zb, zi)),
                                          ((xb, xi), (yb, yi), (za, zz)), ((xa, xx), (yb, yi), (za, zz)),
                                          ((xb, xi), (ya, yy), (za, zz)), ((xa, xx), (ya, yy), (za, zz))]:
        wgt = ((wx * wy).astype(F) * wz).astype(F)
        tS, tW = (wgt * S[ix, iy, iz]).astype(F), (wgt * W[ix, iy, iz]).astype(F)
        accS = tS if accS is None else (accS + tS).astype(F)
        accW = tW if accW is None else (accW + tW).astype(F)
    out = np.full((h, w), np.nan, dtype=F)
    with np.errstate(invalid="ignore", divide="ignore"):
        out[valid] = (accS / accW).astype(F)
    return out


def backproject(z, zf, K):
    h, w = z.shape
    fx, fy, cx, cy = F(K[0, 0]), F(K[1, 1]), F(K[0, 2]), F(K[1, 2])
    u, v = np.meshgrid(np.arange(w, dtype=F), np.arange(h, dtype=F))
    xyz = np.empty((h, w, 3), dtype=F)
    xyz[..., 0] = ((u - cx) / fx).astype(F) * z
    xyz[..., 1] = ((v - cy) / fy).astype(F) * z
    xyz[..., 2] = zf
    bad = np.isnan(z) | np.isnan(zf)
    xyz[bad] = np.nan
    return xyz


def normals(xyz, r, max_change):
    h, w, _ = xyz.shape
    dh = np.zeros((h, w, 3), dtype=F)
    dv = np.zeros((h, w, 3), dtype=F)
    ok = np.zeros((h, w), dtype=bool)
    c = xyz[1:-1, 1:-1]
    l, rr = xyz[1:-1, :-2], xyz[1:-1, 2:]
    up, dn = xyz[:-2, 1:-1], xyz[2:, 1:-1]
    with np.errstate(invalid="ignore"):
        valid = ~(np.isnan(c[..., 2]) | np.isnan(l[..., 2]) | np.isnan(rr[..., 2]) | np.isnan(up[..., 2]) | np.isnan(dn[..., 2]))
        lim = (F(2.0) * F(max_change) * c[..., 2]).astype(F)
        valid &= (np.abs(rr[..., 2] - l[..., 2]) <= lim) & (np.abs(dn[..., 2] - up[..., 2]) <= lim)
    dh[1:-1, 1:-1] = np.where(valid[..., None], rr - l, 0).astype(F)
    dv[1:-1, 1:-1] = np.where(valid[..., None], dn - up, 0).astype(F)
    ok[1:-1, 1:-1] = valid
    # windowed sums, separably like the kernel: each row left to right, then the row sums top to bottom
    # (an invalid gradient is stored as zeros, so adding it is the same as skipping it)
    def box(a):
        pad = np.zeros((h + 2 * r, w + 2 * r) + a.shape[2:], dtype=F)
        pad[r:r + h, r:r + w] = a
        rows = np.zeros((h + 2 * r, w) + a.shape[2:], dtype=F)
        for dx in range(2 * r + 1):
            rows = (rows + pad[:, dx:dx + w]).astype(F)
        out = np.zeros((h, w) + a.shape[2:], dtype=F)
        for dy in range(2 * r + 1):
            out = (out + rows[dy:dy + h]).astype(F)
        return out
    sh, sv, cnt = box(dh), box(dv), box(ok.astype(F))
    c0 = (sv[..., 1] * sh[..., 2]).astype(F) - (sv[..., 2] * sh[..., 1]).astype(F)
    c1 = (sv[..., 2] * sh[..., 0]).astype(F) - (sv[..., 0] * sh[..., 2]).astype(F)
    c2 = (sv[..., 0] * sh[..., 1]).astype(F) - (sv[..., 1] * sh[..., 0]).astype(F)
    n = np.stack([c0, c1, c2], -1).astype(F)
    length = np.sqrt(((c0 * c0).astype(F) + (c1 * c1).astype(F)).astype(F) + (c2 * c2).astype(F)).astype(F)
    with np.errstate(invalid="ignore", divide="ignore"):
        n = (n / length[..., None]).astype(F)
        flip = ((n[..., 0] * xyz[..., 0]).astype(F) + (n[..., 1] * xyz[..., 1]).astype(F)).astype(F) + (n[..., 2] * xyz[..., 2]).astype(F) > 0
    n = np.where(flip[..., None], -n, n)
    bad = np.isnan(xyz[..., 2]) | (cnt == 0) | ~(length > 0)
    n[bad] = np.nan
    return n


def preprocess(depth, K, depth_scale=1.0 / 5000.0, sigma_s=15.0, sigma_r=0.05, radius=30, normal_radius=5,
               max_depth_change=0.02, grid_filter=1):
    z = depth_to_z(depth, depth_scale)
    if grid_filter and radius > 0:
        zf = bilateral_grid(z, sigma_s, sigma_r)
    else:
        zf = bilateral(z, radius, sigma_s, sigma_r)
    xyz = backproject(z, zf, K)
    return xyz, normals(xyz, normal_radius, max_depth_change)
