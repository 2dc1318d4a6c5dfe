#!/usr/bin/env python3
"""Resample the fr1/plant ground-truth trajectory that ships with the reference to 30 Hz.

Reads  /root/reference/src/rgbd_dataset_freiburg1_plant-groundtruth.txt  (4124 mocap poses @100 Hz,
`timestamp tx ty tz qx qy qz qw`; a data file, not code) and writes
tracking_sdf_amd/data/fr1_plant_gt_30hz.txt: one pose per 1/30 s (positions linearly interpolated,
quaternions normalised-lerped between the two neighbouring mocap samples), same column layout,
6 decimals.  It drives the synthetic depth sequence and is the
ATE reference.  Run in the build container only (the GPU box has no /root/reference).
"""
import os
import sys

import numpy as np

SRC = "/root/reference/src/rgbd_dataset_freiburg1_plant-groundtruth.txt"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "tracking_sdf_amd", "data", "fr1_plant_gt_30hz.txt")


def main():
    g = np.loadtxt(SRC)
    ts = g[:, 0]
    t_out = np.arange(ts[0], ts[-1], 1.0 / 30.0)
    rows = []
    for t in t_out:
        j = int(np.searchsorted(ts, t, side="right"))
        j = min(max(j, 1), len(ts) - 1)
        a, b = g[j - 1], g[j]
        w = 0.0 if b[0] == a[0] else (t - a[0]) / (b[0] - a[0])
        p = (1 - w) * a[1:4] + w * b[1:4]
        qa, qb = a[4:8], b[4:8]
        if np.dot(qa, qb) < 0:
            qb = -qb
        q = (1 - w) * qa + w * qb
        q /= np.linalg.norm(q)
        rows.append([t, *p, *q])
    rows = np.array(rows)
    os.makedirs(os.path.dirname(DST), exist_ok=True)
    with open(DST, "w") as f:
        f.write("# fr1/plant ground truth resampled to 30 Hz by tools/make_gt_30hz.py\n")
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for r in rows:
            f.write("%.4f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n" % tuple(r))
    print(f"wrote {len(rows)} poses to {DST}")


if __name__ == "__main__":
    sys.exit(main())
