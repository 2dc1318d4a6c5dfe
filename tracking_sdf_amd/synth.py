"""Synthetic RGB-D sequence for the tracking_sdf hot path.

No TUM image data exists in the build container or on the GPU box, so the benchmark and the
parity tests are fed by an analytic scene rendered along the *real* fr1/plant ground-truth camera
path (tracking_sdf_amd/data/fr1_plant_gt_30hz.txt, re-based so that pose 0 is the reference's
hard-coded initial pose, camera_tracking.cpp:5-7).

Scene (world = the reference's default 6 x 6 x 3.5 m volume, origin (-3,-3,-0.5)):
a room whose walls sit 0.2 m inside the volume faces, one sphere, a table-like cuboid and a box.
All three primitive types have closed-form ray intersections, so depth is exact (no sphere
tracing) and the normals are analytic, flipped toward the camera like PCL's.

Outputs per frame are exactly the arrays the reference's two hot calls receive:
xyz (h,w,3) float32 organised cloud with NaN holes, normals (h,w,3) float32, rgb (h,w,3) uint8.
"""
from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fr1_plant_gt_30hz.txt")

ROT_INIT = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, -1.0, 0]])     # camera_tracking.cpp:7 (det = -1)
TRANS_INIT = np.array([0.0, 0.0, 1.0])                              # camera_tracking.cpp:5

ROOM_LO = np.array([-2.8, -2.8, -0.3])
ROOM_HI = np.array([2.8, 2.8, 2.8])
SPHERE_C = np.array([-1.3, -1.9, 0.9])
SPHERE_R = 0.45
BOXES = (  # (lo, hi) solid cuboids
    (np.array([0.9, -2.7, -0.3]), np.array([2.1, -1.6, 0.45])),     # table
    (np.array([-0.35, -2.75, -0.3]), np.array([0.35, -2.35, 1.3])),  # cabinet against the far wall
)


def default_intrinsics(width=640, height=480):
    """fx = fy = 525 * (W/640), principal point at the image centre (ROS default for 640x480)."""
    f = 525.0 * (width / 640.0)
    return np.array([[f, 0.0, (width - 1) / 2.0], [0.0, f, (height - 1) / 2.0], [0.0, 0.0, 1.0]])


def quat_to_rot(q):
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def load_trajectory(n_frames=None, step=1, path=_DATA):
    """Camera-to-world poses (R_k, t_k) along fr1/plant, re-based to the reference's initial pose.

    T_k = T_init o (G_0^-1 G_k): every R_k inherits det = -1 from the reference's mirrored world.
    Returns (timestamps, R (n,3,3), t (n,3)).
    """
    g = np.loadtxt(path)
    g = g[::step]
    if n_frames is not None:
        g = g[:n_frames]
    R0 = quat_to_rot(g[0, 4:8])
    t0 = g[0, 1:4]
    Rs, ts = [], []
    for row in g:
        Rk = quat_to_rot(row[4:8])
        dR = R0.T @ Rk
        dt = R0.T @ (row[1:4] - t0)
        Rs.append(ROT_INIT @ dR)
        ts.append(ROT_INIT @ dt + TRANS_INIT)
    return g[:, 0].copy(), np.array(Rs), np.array(ts)


def _ray_box_inside(o, d, lo, hi):
    """Exit distance of rays that start inside the box, and the inward normal of the face hit."""
    with np.errstate(divide="ignore", invalid="ignore"):
        s_hi = (hi - o) / d
        s_lo = (lo - o) / d
    s = np.where(d > 0, s_hi, np.where(d < 0, s_lo, np.inf))
    ax = np.argmin(s, axis=-1)
    smin = np.take_along_axis(s, ax[..., None], -1)[..., 0]
    n = np.zeros(d.shape)
    sign = -np.sign(np.take_along_axis(d, ax[..., None], -1)[..., 0])
    np.put_along_axis(n, ax[..., None], sign[..., None], -1)
    return smin, n


def _ray_box_outside(o, d, lo, hi):
    with np.errstate(divide="ignore", invalid="ignore"):
        s1 = (lo - o) / d
        s2 = (hi - o) / d
    near = np.minimum(s1, s2)
    far = np.maximum(s1, s2)
    near = np.where(np.isnan(near), -np.inf, near)
    far = np.where(np.isnan(far), np.inf, far)
    ax = np.argmax(near, axis=-1)
    s_near = np.take_along_axis(near, ax[..., None], -1)[..., 0]
    s_far = far.min(axis=-1)
    hit = (s_near < s_far) & (s_near > 1e-9)
    n = np.zeros(d.shape)
    sign = -np.sign(np.take_along_axis(d, ax[..., None], -1)[..., 0])
    np.put_along_axis(n, ax[..., None], sign[..., None], -1)
    return np.where(hit, s_near, np.inf), n


def _ray_sphere(o, d, c, r):
    oc = o - c
    a = (d * d).sum(-1)
    b = 2.0 * (d * oc).sum(-1)
    cc = (oc * oc).sum(-1) - r * r
    disc = b * b - 4 * a * cc
    ok = disc > 0
    sq = np.sqrt(np.where(ok, disc, 0.0))
    s = (-b - sq) / (2 * a)
    hit = ok & (s > 1e-9)
    s = np.where(hit, s, np.inf)
    p = o + np.where(hit, s, 0.0)[..., None] * d
    n = (p - c) / r
    return s, n


def render_frame(R, t, K, width=640, height=480, noise=False, holes=0.0, max_depth=5.0, min_depth=0.4,
                 rng=None):
    """Render one organised cloud from camera-to-world pose (R, t).

    noise: Kinect-like axial noise sigma(z) = 0.0012 + 0.0019 (z - 0.4)^2 metres; holes: fraction of
    random NaN pixels; pixels outside [min_depth, max_depth] are NaN (sensor range).
    """
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1)   # z = 1
    dw = dc @ R.T
    o = np.broadcast_to(t, dw.shape)
    best, nrm = _ray_box_inside(o, dw, ROOM_LO, ROOM_HI)
    prim = np.zeros(best.shape, dtype=np.int32)
    s, n = _ray_sphere(o, dw, SPHERE_C, SPHERE_R)
    m = s < best
    best = np.where(m, s, best); nrm = np.where(m[..., None], n, nrm); prim = np.where(m, 1, prim)
    for bi, (lo, hi) in enumerate(BOXES):
        s, n = _ray_box_outside(o, dw, lo, hi)
        m = s < best
        best = np.where(m, s, best); nrm = np.where(m[..., None], n, nrm); prim = np.where(m, 2 + bi, prim)
    z = best.copy()
    pw = o + z[..., None] * dw                       # exact world hit point (for the texture)
    if noise:
        if rng is None:
            rng = np.random.default_rng(0)
        z = z + rng.standard_normal(z.shape) * (0.0012 + 0.0019 * (z - 0.4) ** 2)
    valid = np.isfinite(z) & (z >= min_depth) & (z <= max_depth)
    if holes > 0:
        if rng is None:
            rng = np.random.default_rng(0)
        valid &= rng.random(z.shape) >= holes
    xyz = (z[..., None] * dc).astype(np.float32)
    n_cam = nrm @ R                                   # R^-1 n = R^T n  (R orthogonal, det +-1)
    flip = (n_cam * dc).sum(-1) > 0                   # make the normal face the camera
    n_cam = np.where(flip[..., None], -n_cam, n_cam).astype(np.float32)
    xyz[~valid] = np.nan
    n_cam[~valid] = np.nan
    # procedural colour: per-primitive base colour x 0.25 m world checker
    base = np.array([[200, 200, 190], [60, 160, 70], [170, 110, 60], [90, 90, 170]], dtype=np.float64)[prim]
    chk = (np.floor(pw[..., 0] * 4) + np.floor(pw[..., 1] * 4) + np.floor(pw[..., 2] * 4)) % 2
    rgb = np.clip(base * (0.75 + 0.25 * chk[..., None]), 0, 255).astype(np.uint8)
    rgb[~valid] = 0
    return xyz, n_cam, rgb


class Sequence:
    """The synthetic stand-in for a TUM sequence: frames rendered on demand along fr1/plant."""

    def __init__(self, n_frames=100, width=640, height=480, noise=False, holes=0.0, seed=0, step=1, K=None):
        self.width, self.height = int(width), int(height)
        self.K = default_intrinsics(width, height) if K is None else np.asarray(K, dtype=np.float64)
        self.stamps, self.R, self.t = load_trajectory(n_frames, step)
        self.noise, self.holes, self.seed = bool(noise), float(holes), int(seed)

    def __len__(self):
        return len(self.stamps)

    def frame(self, k):
        rng = np.random.default_rng([self.seed, k])
        return render_frame(self.R[k], self.t[k], self.K, self.width, self.height, self.noise, self.holes, rng=rng)
