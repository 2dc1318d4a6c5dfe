"""NumPy statement of this repository's depth pre-processing stand-in (the parity target of
tracking_sdf_amd/csrc/preproc_kernels.hip).  TEST INFRASTRUCTURE ONLY.

The reference pre-processes with PCL (sdf_reconstruction.cpp:29-49), which is neither in the reference tree
nor in this image: PARITY WITH PCL IS UNPINNED.  Every float operation below is done in float32, in the same
order as the kernels (taps row by row), so the two agree to the last bits except for expf.
"""
import numpy as np

F = np.float32


def depth_to_z(depth, scale):
    if depth.dtype == np.uint16:
        z = depth.astype(F) * F(scale)
        z[depth == 0] = np.nan
    else:
        z = depth.astype(F).copy()
        z[~(z > 0)] = np.nan
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
    # windowed sums, taps row by row (same order as the kernel)
    pdh = np.zeros((h + 2 * r, w + 2 * r, 3), dtype=F); pdh[r:r + h, r:r + w] = dh
    pdv = np.zeros((h + 2 * r, w + 2 * r, 3), dtype=F); pdv[r:r + h, r:r + w] = dv
    pok = np.zeros((h + 2 * r, w + 2 * r), dtype=bool); pok[r:r + h, r:r + w] = ok
    sh = np.zeros((h, w, 3), dtype=F); sv = np.zeros((h, w, 3), dtype=F); cnt = np.zeros((h, w), dtype=F)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            m = pok[r + dy:r + dy + h, r + dx:r + dx + w]
            sh = np.where(m[..., None], (sh + pdh[r + dy:r + dy + h, r + dx:r + dx + w]).astype(F), sh)
            sv = np.where(m[..., None], (sv + pdv[r + dy:r + dy + h, r + dx:r + dx + w]).astype(F), sv)
            cnt = cnt + m.astype(F)
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
               max_depth_change=0.02):
    z = depth_to_z(depth, depth_scale)
    zf = bilateral(z, radius, sigma_s, sigma_r)
    xyz = backproject(z, zf, K)
    return xyz, normals(xyz, normal_radius, max_depth_change)
