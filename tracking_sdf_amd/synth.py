"""Synthetic RGB-D sequence for the tracking_sdf hot path.

No TUM image data exists in the build container or on the GPU box, so the benchmark and the
parity tests are fed by an analytic scene rendered along the *real* fr1/plant ground-truth camera
path (tracking_sdf_amd/data/fr1_plant_gt_30hz.txt, re-based so that pose 0 is the reference's
hard-coded initial pose, camera_tracking.cpp:5-7).

Scene (world = the reference's default 6 x 6 x 3.5 m volume, origin (-3,-3,-0.5)).  fr1/plant is a hand-held
camera circling a potted plant at 0.6-1.0 m; the ground-truth rays meet within 18 cm (mean) of the world point
FOCUS below, so that is where the synthetic plant stands:
  * the plant: a pedestal (cuboid), a pot (vertical cylinder) and foliage (a cluster of spheres) around FOCUS,
    all within 0.3 m of the vertical through it (the camera never comes closer than 0.63 m to FOCUS);
  * the room: walls 0.2 m inside the volume faces, made non-planar by pillars (vertical cylinders) and domes
    (spheres half sunk into walls and floor);
  * clutter along the walls: a table, a cabinet, boxes, a large ball.
(The round-1 scene was an almost empty room -- nothing stood where the camera looks, so the tracker saw
sliding planes at 2-4 m; at 256^3 it lost the camera in the fastest part of the path.)
Every primitive (cuboid, sphere, vertical cylinder) has a closed-form ray intersection, so depth is exact (no
sphere tracing) and the normals are analytic, flipped toward the camera like PCL's.

Outputs per frame are exactly the arrays the reference's two hot calls receive:
xyz (h,w,3) float32 organised cloud with NaN holes, normals (h,w,3) float32, rgb (h,w,3) uint8.

Two renderers of the same scene: render_frame (NumPy, the one the parity tests and fixtures use) and
render_frame_torch (the same arithmetic as torch ops, so that a 1246-frame sequence can be rendered on the GPU
in seconds for the full-sequence ATE; its noise comes from a torch generator, i.e. other random numbers).
"""
from __future__ import annotations

import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fr1_plant_gt_30hz.txt")

ROT_INIT = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, -1.0, 0]])     # camera_tracking.cpp:7 (det = -1)
TRANS_INIT = np.array([0.0, 0.0, 1.0])                              # camera_tracking.cpp:5

ROOM_LO = np.array([-2.8, -2.8, -0.3])
ROOM_HI = np.array([2.8, 2.8, 2.8])
FOCUS = np.array([0.13, -0.90, 0.91])     # least-squares meeting point of the ground-truth viewing rays

# (centre, radius, colour id)
SPHERES = (
    # foliage around FOCUS (top below 1.1 m: the camera passes right above the plant at 1.5 m)
    ((0.13, -0.90, 0.84), 0.15, 1), ((0.27, -0.84, 0.76), 0.10, 1), ((0.00, -0.97, 0.78), 0.11, 1),
    ((0.17, -1.04, 0.88), 0.09, 1), ((0.07, -0.78, 0.90), 0.10, 1), ((0.24, -0.98, 0.96), 0.08, 1),
    ((0.02, -0.86, 0.98), 0.08, 1), ((0.15, -0.88, 1.01), 0.07, 1), ((0.30, -0.76, 0.90), 0.07, 1),
    ((-0.05, -1.05, 0.92), 0.07, 1), ((0.20, -0.72, 0.72), 0.06, 1), ((0.05, -1.08, 0.70), 0.06, 1),
    # the round-1 ball, domes sunk into walls / floor / ceiling
    ((-1.3, -1.9, 0.9), 0.45, 4),
    ((2.95, 0.4, 1.1), 0.60, 5), ((-2.95, -0.8, 1.5), 0.70, 5), ((0.6, 2.95, 0.9), 0.75, 5),
    ((-0.9, -3.0, 1.9), 0.65, 5), ((1.6, -0.2, -0.55), 0.55, 5), ((-1.5, 0.9, -0.6), 0.60, 5),
    ((0.2, 0.6, 3.1), 0.70, 5), ((-1.0, -1.6, 3.05), 0.60, 5),
)
# (lo, hi, colour id) solid cuboids
BOXES = (
    ((-0.07, -1.10, -0.3), (0.33, -0.70, 0.35), 2),      # pedestal under the pot
    ((0.9, -2.7, -0.3), (2.1, -1.6, 0.45), 2),           # table
    ((-0.35, -2.75, -0.3), (0.35, -2.35, 1.3), 3),       # cabinet against the far wall
    ((-2.75, -0.3, -0.3), (-2.2, 0.9, 0.8), 3),          # chest against the -x wall
    ((1.9, 0.8, -0.3), (2.75, 2.0, 1.7), 2),             # wardrobe in the +x +y corner
    ((-1.9, 1.9, -0.3), (-0.7, 2.75, 0.6), 3),           # bench against the +y wall
    ((1.2, -2.4, 0.45), (1.6, -2.0, 0.8), 4),            # box on the table
)
# vertical cylinders (cx, cy, radius, z0, z1, colour id)
CYLINDERS = (
    (0.13, -0.90, 0.13, 0.35, 0.62, 3),                  # pot
    (-2.8, -2.8, 0.5, -0.3, 2.8, 5), (2.8, -2.8, 0.45, -0.3, 2.8, 5),      # pillars in the corners
    (-2.8, 2.8, 0.5, -0.3, 2.8, 5), (2.8, 2.8, 0.4, -0.3, 2.8, 5),
    (2.8, -0.9, 0.3, -0.3, 2.8, 5), (-0.2, 2.8, 0.35, -0.3, 2.8, 5),       # half-pillars on two walls
    (-2.0, -2.8, 0.3, -0.3, 2.8, 5),
    (-1.9, -0.4, 0.18, -0.3, 1.1, 4), (1.1, 1.0, 0.22, -0.3, 0.9, 4),      # free-standing posts
)
# The round-1 benchmark scene (an almost empty room: one ball, a table, a cabinet), kept as a fixed yardstick: kernel
# timings of different rounds are only comparable on the same workload (Sequence(scene="room"), bench.py's
# `round1_scene` leg).  The tracker loses the camera on it at 256^3 -- it is not a tracking benchmark.
ROOM_SCENE = {
    "spheres": (((-1.3, -1.9, 0.9), 0.45, 1),),
    "boxes": (((0.9, -2.7, -0.3), (2.1, -1.6, 0.45), 2), ((-0.35, -2.75, -0.3), (0.35, -2.35, 1.3), 3)),
    "cylinders": (),
}
PLANT_SCENE = {"spheres": SPHERES, "boxes": BOXES, "cylinders": CYLINDERS}
SCENES = {"plant": PLANT_SCENE, "room": ROOM_SCENE}

BASE_COLOURS = np.array([[200, 200, 190], [60, 160, 70], [170, 110, 60], [90, 90, 170], [180, 70, 70],
                         [150, 150, 165]], dtype=np.float64)


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


# ---------------------------------------------------------------------------------------------------------
# NumPy renderer

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


def _ray_cylinder(o, d, cx, cy, r, z0, z1):
    """Solid vertical cylinder = (infinite cylinder) AND (slab z0..z1): entry = the later of the two entries."""
    ox, oy = o[..., 0] - cx, o[..., 1] - cy
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    a = dx * dx + dy * dy
    b = 2.0 * (ox * dx + oy * dy)
    cc = ox * ox + oy * oy - r * r
    disc = b * b - 4 * a * cc
    ok = (disc > 0) & (a > 1e-18)
    sq = np.sqrt(np.where(ok, disc, 0.0))
    a2 = np.where(ok, 2 * a, 1.0)
    side_in = np.where(ok, (-b - sq) / a2, np.inf)
    side_out = np.where(ok, (-b + sq) / a2, -np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        sa = (z0 - o[..., 2]) / dz
        sb = (z1 - o[..., 2]) / dz
    cap_in = np.minimum(sa, sb)
    cap_out = np.maximum(sa, sb)
    cap_in = np.where(np.isnan(cap_in), -np.inf, cap_in)
    cap_out = np.where(np.isnan(cap_out), np.inf, cap_out)
    s_in = np.maximum(side_in, cap_in)
    s_out = np.minimum(side_out, cap_out)
    hit = (s_in < s_out) & (s_in > 1e-9)
    s = np.where(hit, s_in, np.inf)
    sh = np.where(hit, s_in, 0.0)
    from_side = side_in >= cap_in
    n = np.zeros(d.shape)
    n[..., 0] = np.where(from_side, (ox + sh * dx) / r, 0.0)
    n[..., 1] = np.where(from_side, (oy + sh * dy) / r, 0.0)
    n[..., 2] = np.where(from_side, 0.0, -np.sign(dz))
    return s, n


def render_frame(R, t, K, width=640, height=480, noise=False, holes=0.0, max_depth=5.0, min_depth=0.4,
                 rng=None, scene="plant"):
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

    def take(s, n, colour):
        nonlocal best, nrm, prim
        m = s < best
        best = np.where(m, s, best); nrm = np.where(m[..., None], n, nrm); prim = np.where(m, colour, prim)
    sc = SCENES[scene]
    for c, r, colour in sc["spheres"]:
        take(*_ray_sphere(o, dw, np.asarray(c), r), colour)
    for lo, hi, colour in sc["boxes"]:
        take(*_ray_box_outside(o, dw, np.asarray(lo), np.asarray(hi)), colour)
    for cx, cy, r, z0, z1, colour in sc["cylinders"]:
        take(*_ray_cylinder(o, dw, cx, cy, r, z0, z1), colour)
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
    base = BASE_COLOURS[prim]
    chk = (np.floor(pw[..., 0] * 4) + np.floor(pw[..., 1] * 4) + np.floor(pw[..., 2] * 4)) % 2
    rgb = np.clip(base * (0.75 + 0.25 * chk[..., None]), 0, 255).astype(np.uint8)
    rgb[~valid] = 0
    return xyz, n_cam, rgb


# ---------------------------------------------------------------------------------------------------------
# torch renderer (same arithmetic, f64, any device): all primitives of a kind are intersected at once

class TorchRenderer:
    """render(R, t) -> (xyz, nrm, rgb) torch tensors on `device`, shaped like render_frame's arrays."""

    def __init__(self, K, width=640, height=480, device="cpu", max_depth=5.0, min_depth=0.4, scene="plant"):
        import torch
        sc = SCENES[scene]
        SPHERES, BOXES = sc["spheres"], sc["boxes"]
        CYLINDERS = sc["cylinders"] or ((0.0, 0.0, 0.01, -9.0, -8.9, 0),)      # a degenerate one far below the room keeps the tensor shapes
        self.torch = torch
        self.dev = torch.device(device)
        f64 = dict(dtype=torch.float64, device=self.dev)
        K = np.asarray(K, dtype=np.float64)
        v, u = torch.meshgrid(torch.arange(height, **f64), torch.arange(width, **f64), indexing="ij")
        self.dc = torch.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)], -1)
        self.max_depth, self.min_depth = max_depth, min_depth
        self.room_lo = torch.tensor(ROOM_LO, **f64)
        self.room_hi = torch.tensor(ROOM_HI, **f64)
        self.sph_c = torch.tensor([s[0] for s in SPHERES], **f64)
        self.sph_r = torch.tensor([s[1] for s in SPHERES], **f64)
        self.box_lo = torch.tensor([b[0] for b in BOXES], **f64)
        self.box_hi = torch.tensor([b[1] for b in BOXES], **f64)
        self.cyl = torch.tensor([c[:5] for c in CYLINDERS], **f64)
        # colour ids in the order the NumPy renderer tests the primitives (ties go to the earlier one)
        ids = [0] + [s[2] for s in SPHERES] + [b[2] for b in BOXES] + [c[5] for c in CYLINDERS]
        self.colour_id = torch.tensor(ids, dtype=torch.long, device=self.dev)
        self.base = torch.tensor(BASE_COLOURS, **f64)

    def render(self, R, t, noise=False, holes=0.0, generator=None):
        torch = self.torch
        f64 = dict(dtype=torch.float64, device=self.dev)
        inf = float("inf")
        R = torch.as_tensor(np.asarray(R, dtype=np.float64), **f64)
        o = torch.as_tensor(np.asarray(t, dtype=np.float64), **f64)
        dc = self.dc
        d = dc @ R.T                                             # (h,w,3)
        # room (rays start inside): exit through the nearest face
        s_hi = (self.room_hi - o) / d
        s_lo = (self.room_lo - o) / d
        s = torch.where(d > 0, s_hi, torch.where(d < 0, s_lo, torch.full_like(d, inf)))
        s_room, ax = s.min(dim=-1)
        n_room = torch.zeros_like(d).scatter_(-1, ax[..., None], -torch.sign(d.gather(-1, ax[..., None])))
        # spheres: (h,w,S)
        oc = o - self.sph_c                                      # (S,3)
        a = (d * d).sum(-1)[..., None]
        b = 2.0 * (d @ oc.T)
        cc = (oc * oc).sum(-1) - self.sph_r ** 2
        disc = b * b - 4 * a * cc
        ok = disc > 0
        sq = torch.sqrt(torch.where(ok, disc, torch.zeros_like(disc)))
        s_sph = (-b - sq) / (2 * a)
        s_sph = torch.where(ok & (s_sph > 1e-9), s_sph, torch.full_like(s_sph, inf))
        # boxes (outside): (h,w,B,3) slabs
        dd = d[..., None, :]
        s1 = (self.box_lo - o) / dd
        s2 = (self.box_hi - o) / dd
        near = torch.minimum(s1, s2)
        far = torch.maximum(s1, s2)
        near = torch.where(torch.isnan(near), torch.full_like(near, -inf), near)
        far = torch.where(torch.isnan(far), torch.full_like(far, inf), far)
        s_near, bax = near.max(dim=-1)
        s_far = far.min(dim=-1).values
        s_box = torch.where((s_near < s_far) & (s_near > 1e-9), s_near, torch.full_like(s_near, inf))
        # cylinders: (h,w,C)
        cx, cy, cr, z0, z1 = (self.cyl[:, q] for q in range(5))
        ox, oy = o[0] - cx, o[1] - cy
        dx, dy, dz = d[..., 0:1], d[..., 1:2], d[..., 2:3]
        ca = dx * dx + dy * dy
        cb = 2.0 * (ox * dx + oy * dy)
        ccc = ox * ox + oy * oy - cr * cr
        cdisc = cb * cb - 4 * ca * ccc
        cok = (cdisc > 0) & (ca > 1e-18)
        csq = torch.sqrt(torch.where(cok, cdisc, torch.zeros_like(cdisc)))
        a2 = torch.where(cok, 2 * ca, torch.ones_like(cdisc))
        side_in = torch.where(cok, (-cb - csq) / a2, torch.full_like(cdisc, inf))
        side_out = torch.where(cok, (-cb + csq) / a2, torch.full_like(cdisc, -inf))
        sa = (z0 - o[2]) / dz
        sb = (z1 - o[2]) / dz
        cap_in = torch.minimum(sa, sb)
        cap_out = torch.maximum(sa, sb)
        cap_in = torch.where(torch.isnan(cap_in), torch.full_like(cap_in, -inf), cap_in)
        cap_out = torch.where(torch.isnan(cap_out), torch.full_like(cap_out, inf), cap_out)
        c_in = torch.maximum(side_in, cap_in)
        c_out = torch.minimum(side_out, cap_out)
        s_cyl = torch.where((c_in < c_out) & (c_in > 1e-9), c_in, torch.full_like(c_in, inf))
        # nearest primitive; argmin returns the first minimum = the NumPy renderer's strict "<" update order
        all_s = torch.cat([s_room[..., None], s_sph, s_box, s_cyl], -1)
        z, which = all_s.min(dim=-1)
        nS, nB = s_sph.shape[-1], s_box.shape[-1]
        pw = o + z[..., None] * d
        # normal of the winner
        nrm = n_room
        is_s = (which >= 1) & (which < 1 + nS)
        if True:
            si = (which - 1).clamp(0, nS - 1)
            n_s = (pw - self.sph_c[si]) / self.sph_r[si][..., None]
            nrm = torch.where(is_s[..., None], n_s, nrm)
        is_b = (which >= 1 + nS) & (which < 1 + nS + nB)
        if True:
            bi = (which - 1 - nS).clamp(0, nB - 1)
            axb = bax.gather(-1, bi[..., None])                               # (h,w,1) entry axis of that box
            n_b = torch.zeros_like(d).scatter_(-1, axb, -torch.sign(d.gather(-1, axb)))
            nrm = torch.where(is_b[..., None], n_b, nrm)
        is_c = which >= 1 + nS + nB
        if True:
            ci = (which - 1 - nS - nB).clamp(0, s_cyl.shape[-1] - 1)
            from_side = (side_in >= cap_in).gather(-1, ci[..., None])[..., 0]
            n_c = torch.zeros_like(d)
            n_c[..., 0] = torch.where(from_side, (pw[..., 0] - cx[ci]) / cr[ci], torch.zeros_like(z))
            n_c[..., 1] = torch.where(from_side, (pw[..., 1] - cy[ci]) / cr[ci], torch.zeros_like(z))
            n_c[..., 2] = torch.where(from_side, torch.zeros_like(z), -torch.sign(d[..., 2]))
            nrm = torch.where(is_c[..., None], n_c, nrm)
        zz = z
        if noise:
            zz = z + torch.randn(z.shape, generator=generator, **f64) * (0.0012 + 0.0019 * (z - 0.4) ** 2)
        valid = torch.isfinite(zz) & (zz >= self.min_depth) & (zz <= self.max_depth)
        if holes > 0:
            valid &= torch.rand(z.shape, generator=generator, **f64) >= holes
        xyz = (zz[..., None] * dc).to(torch.float32)
        n_cam = nrm @ R
        flip = (n_cam * dc).sum(-1) > 0
        n_cam = torch.where(flip[..., None], -n_cam, n_cam).to(torch.float32)
        nan = float("nan")
        xyz = torch.where(valid[..., None], xyz, torch.full_like(xyz, nan))
        n_cam = torch.where(valid[..., None], n_cam, torch.full_like(n_cam, nan))
        base = self.base[self.colour_id[which]]
        chk = (torch.floor(pw[..., 0] * 4) + torch.floor(pw[..., 1] * 4) + torch.floor(pw[..., 2] * 4)) % 2
        rgb = torch.clamp(base * (0.75 + 0.25 * chk[..., None]), 0, 255).to(torch.uint8)
        rgb = torch.where(valid[..., None], rgb, torch.zeros_like(rgb))
        return xyz.contiguous(), n_cam.contiguous(), rgb.contiguous()


class Sequence:
    """The synthetic stand-in for a TUM sequence: frames rendered on demand along fr1/plant."""

    def __init__(self, n_frames=100, width=640, height=480, noise=False, holes=0.0, seed=0, step=1, K=None, scene="plant"):
        self.scene = scene
        self.width, self.height = int(width), int(height)
        self.K = default_intrinsics(width, height) if K is None else np.asarray(K, dtype=np.float64)
        self.stamps, self.R, self.t = load_trajectory(n_frames, step)
        self.noise, self.holes, self.seed = bool(noise), float(holes), int(seed)
        self._tr = None

    def __len__(self):
        return len(self.stamps)

    def frame(self, k):
        rng = np.random.default_rng([self.seed, k])
        return render_frame(self.R[k], self.t[k], self.K, self.width, self.height, self.noise, self.holes, rng=rng,
                            scene=self.scene)

    def frame_torch(self, k, device="cuda"):
        """Frame k rendered with torch on `device` (device tensors; the noise differs from frame(k)'s)."""
        import torch
        if self._tr is None or str(self._tr.dev) != str(torch.device(device)):
            self._tr = TorchRenderer(self.K, self.width, self.height, device, scene=self.scene)
            self._gen = torch.Generator(device=self._tr.dev)
        self._gen.manual_seed(self.seed * 1000003 + k)
        return self._tr.render(self.R[k], self.t[k], self.noise, self.holes, self._gen)
