#!/usr/bin/env python3
"""CPU statistics of the integrate work list for one frame of the bench stream (no GPU): how many lanes of the listed
64-voxel items are inside the row's frustum interval, how many are updated, and what items of 32 or 16 voxels would list.
Approximate arithmetic (f64 NumPy, not the reference's bit-exact mix): counts only."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tracking_sdf_amd import synth

m = 512
frame = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seq = synth.Sequence(n_frames=frame + 1, width=640, height=480, noise=True, holes=0.02)
xyz, nrm, rgb = seq.frame(frame)
R, t, K = seq.R[frame], seq.t[frame], seq.K
# bench re-bases the path to the reference's initial pose; for statistics the raw pose in a volume centred on the scene will do
W, H = 640, 480
ext = np.array([6.0, 6.0, 3.5]); org = np.array([-3.0, -3.0, -0.5])
from tracking_sdf_amd import synth as _s
delta, eps = 0.3, 0.025
Rinv = R.T; tinv = -R.T @ t
P = xyz.reshape(H, W, 3); N = nrm.reshape(H, W, 3)
valid_px = ~(np.isnan(P[..., 0]) | np.isnan(P[..., 1]) | np.isnan(N[..., 0]) | np.isnan(N[..., 1]) | np.isnan(N[..., 2]))
cell = ext / m
gz = org[2] + cell[2] * (np.arange(m) + 0.5)
tot = {g: 0 for g in (64, 32, 16)}
livechunks = {g: 0 for g in (64, 32, 16)}        # chunks with at least one updated voxel: what a perfect cull would list
n_geom = n_live = n_rows = 0
runs = []
for i in range(m):
    gx = org[0] + cell[0] * (i + 0.5)
    gy = org[1] + cell[1] * (np.arange(m) + 0.5)
    G = np.stack(np.broadcast_arrays(gx, gy[:, None], gz[None, :]), -1)          # (j, k, 3)
    pc = G @ Rinv.T + tinv
    z = pc[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (K[0, 0] * pc[..., 0] + K[0, 2] * z) / z
        v = (K[1, 1] * pc[..., 1] + K[1, 2] * z) / z
    geom = (z >= 0) & (u > -1) & (u < W) & (v > -1) & (v < H)
    ui = np.clip(np.trunc(np.where(geom, u, 0)).astype(int), 0, W - 1); vi = np.clip(np.trunc(np.where(geom, v, 0)).astype(int), 0, H - 1)
    ok = geom & valid_px[vi, ui]
    d = np.einsum("jkc,jkc->jk", P[vi, ui] - pc, N[vi, ui])
    live = ok & (d <= delta)
    rows = geom.any(1)
    k0 = np.where(rows, geom.argmax(1), 0); k1 = np.where(rows, m - 1 - geom[:, ::-1].argmax(1), -1)
    k0w = np.maximum(k0 - 1, 0); k1w = np.minimum(k1 + 1, m - 1)
    for g in tot:
        tot[g] += int((rows * ((k1w // g) - (k0w // g) + 1) * g).sum())
    for g in livechunks:
        livechunks[g] += int(live.reshape(m, m // g, g).any(2).sum())
    n_geom += int(geom.sum()); n_live += int(live.sum()); n_rows += int(rows.sum())
print({"rows_with_items": n_rows, "lanes_in_frustum": n_geom, "updated": n_live,
       "listed_lanes_by_item_size": tot, "items": {g: tot[g] // g for g in tot},
       "chunks_with_an_updated_voxel": livechunks, "lanes_of_those": {g: livechunks[g] * g for g in livechunks},
       "updated_fraction_by_item_size": {g: round(n_live / tot[g], 3) for g in tot}})
