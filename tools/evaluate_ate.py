#!/usr/bin/env python3
"""Absolute trajectory error (ATE-RMSE) between two TUM-format trajectories
(`timestamp tx ty tz qx qy qz qw` per line, '#' comments).

Poses are associated by nearest timestamp (max difference --max-dt), the estimate is aligned to the ground
truth with Horn's closed-form least-squares similarity-free alignment (rotation + translation).  The
reference's world frame is a MIRROR image of the mocap frame (its initial "rotation" has det = -1,
camera_tracking.cpp:7), which no proper rotation can undo, so by default the alignment may include a
reflection (--proper-only switches that off); the chosen determinant is reported.
"""
import argparse
import json
import sys

import numpy as np


def read_tum(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            v = [float(x) for x in line.replace(",", " ").split()]
            if len(v) >= 4:
                rows.append(v[:4])
    a = np.array(rows)
    return a[:, 0], a[:, 1:4]


def associate(t_est, t_gt, max_dt):
    idx = np.searchsorted(t_gt, t_est)
    idx = np.clip(idx, 1, len(t_gt) - 1)
    left = np.abs(t_gt[idx - 1] - t_est) <= np.abs(t_gt[idx] - t_est)
    j = np.where(left, idx - 1, idx)
    ok = np.abs(t_gt[j] - t_est) <= max_dt
    return np.nonzero(ok)[0], j[ok]


def align(est, gt, allow_reflection=True):
    """Least-squares R (orthogonal), t with R est + t ~ gt.  Returns aligned est, R, t."""
    ce, cg = est.mean(0), gt.mean(0)
    H = (est - ce).T @ (gt - cg)
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0 and not allow_reflection:
        S = np.diag([1.0, 1.0, -1.0])
        R = Vt.T @ S @ U.T
    t = cg - R @ ce
    return est @ R.T + t, R, t


def ate(est_path, gt_path, max_dt=0.02, allow_reflection=True):
    te, pe = read_tum(est_path)
    tg, pg = read_tum(gt_path)
    order = np.argsort(tg)
    tg, pg = tg[order], pg[order]
    ie, ig = associate(te, tg, max_dt)
    if len(ie) < 3:
        raise SystemExit("fewer than 3 associated poses")
    al, R, t = align(pe[ie], pg[ig], allow_reflection)
    err = np.linalg.norm(al - pg[ig], axis=1)
    return {"pairs": int(len(ie)), "ate_rmse_m": float(np.sqrt(np.mean(err ** 2))), "ate_mean_m": float(err.mean()),
            "ate_median_m": float(np.median(err)), "ate_max_m": float(err.max()),
            "alignment_det": float(np.linalg.det(R)), "reflection_allowed": bool(allow_reflection)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("estimate")
    ap.add_argument("ground_truth")
    ap.add_argument("--max-dt", type=float, default=0.02)
    ap.add_argument("--proper-only", action="store_true")
    a = ap.parse_args()
    print(json.dumps(ate(a.estimate, a.ground_truth, a.max_dt, not a.proper_only)))


if __name__ == "__main__":
    sys.exit(main())
