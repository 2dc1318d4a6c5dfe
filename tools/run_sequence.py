#!/usr/bin/env python3
"""Offline frame driver: the reference's kinect_callback sequencing (sdf_reconstruction.cpp:21-80) over a
whole sequence on the HIP path, writing ./trajectory.txt in the reference's format (:4-17).

  frame 1: integrate at the initial pose; frame k>1: track -> append pose -> integrate   (:69-74)
  --ground-truth-poses: the reference's _useGroundTruth switch (:51-66): no tracking, fuse at the given poses

Input: the synthetic fr1/plant stream (default; no TUM images exist on the box), or --tum DIR with a TUM
RGB-D directory (depth.txt + depth/*.png, 16-bit, /5000 m): the raw depth image goes to tsdf_set_depth_frame,
i.e. back-projection (--fx/--fy/--cx/--cy), bilateral filter and normals run on the GPU (the reference uses
PCL's bilateral filter + integral image normals, which are not available here: parity of that pre-processing
is UNPINNED).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def quat_from_rot(R):
    """Eigen::Quaterniond(Matrix3d) (what writePoseToFile does); meaningless for det = -1 but reproduced."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        s = np.sqrt(t + 1.0)
        q[3] = 0.5 * s
        s = 0.5 / s
        q[0], q[1], q[2] = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(max(R[i, i] - R[j, j] - R[k, k] + 1.0, 0.0))
        q[i] = 0.5 * s
        s = 0.5 / s if s > 0 else 0.0
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    return q


def tum_frames(root, limit):
    """Yields (stamp, depth uint16 image) of a TUM RGB-D directory."""
    from PIL import Image
    items = []
    with open(os.path.join(root, "depth.txt")) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            ts_, name = line.split()[:2]
            items.append((float(ts_), name))
    if limit:
        items = items[:limit]
    for stamp, name in items:
        yield stamp, np.asarray(Image.open(os.path.join(root, name))).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", dest="m", type=int, default=256)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--tum", default=None)
    ap.add_argument("--fx", type=float, default=525.0)
    ap.add_argument("--fy", type=float, default=525.0)
    ap.add_argument("--cx", type=float, default=319.5)
    ap.add_argument("--cy", type=float, default=239.5)
    ap.add_argument("--ground-truth-poses", action="store_true")
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--out", default="trajectory.txt")
    ap.add_argument("--gt-out", default=None, help="also write the synthetic ground truth (TUM format)")
    ap.add_argument("--save-volume", default=None)
    a = ap.parse_args()

    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth
    sdf = ts.SDF(a.m, with_color=not a.tum)        # TUM depth-only input carries no registered colour here
    trk = ts.CameraTracking(sdf=sdf)
    gt_R = gt_t = None
    if a.tum:
        K = np.array([[a.fx, 0, a.cx], [0, a.fy, a.cy], [0, 0, 1.0]])
        stream = ((st_, d16, None, None) for st_, d16 in tum_frames(a.tum, a.frames))
    else:
        seq = synth.Sequence(n_frames=a.frames, width=a.width, height=a.height, noise=not a.no_noise,
                             holes=0.0 if a.no_noise else 0.02)
        K, gt_R, gt_t = seq.K, seq.R, seq.t
        stream = ((seq.stamps[k],) + seq.frame(k) for k in range(len(seq)))
    trk.set_K(K)
    open(a.out, "w").close()                                   # the reference appends to ./trajectory.txt
    n = errs = 0
    iters = []
    t_hot = 0.0
    for frame_num, (stamp, xyz, nrm, rgb) in enumerate(stream, start=1):
        t0 = time.perf_counter()
        if a.tum:
            sdf.set_depth_frame(xyz, None)                      # `xyz` holds the uint16 depth image here
            xyz = None
        if a.ground_truth_poses and gt_R is not None:
            trk.set_camera_transformation(gt_R[frame_num - 1], gt_t[frame_num - 1])
        elif frame_num > 1:
            try:
                st = trk.estimate_new_position(sdf, xyz)
                iters.append(st["iterations"])
            except ts.TsdfError as e:                           # the reference would carry a NaN pose on
                errs += 1
                print(f"frame {frame_num}: {e}", file=sys.stderr)
            q = quat_from_rot(trk.rot)
            with open(a.out, "a") as f:
                f.write("%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n" % (stamp, *trk.trans, *q))
        if a.tum:
            sdf.update(want_stats=False)
        else:
            sdf.update(trk, xyz, nrm, rgb, want_stats=False)
        sdf.synchronize()
        t_hot += time.perf_counter() - t0
        n += 1
    out = {"frames": n, "track_errors": errs, "mean_gn_iterations": float(np.mean(iters)) if iters else None,
           "hot_path_fps_incl_host_copies": n / t_hot, "trajectory": a.out}
    if gt_t is not None and not a.ground_truth_poses:
        gt_path = a.gt_out or (a.out + ".gt")
        with open(gt_path, "w") as f:
            for k in range(n):
                f.write("%.4f %.6f %.6f %.6f 0 0 0 1\n" % (seq.stamps[k], *gt_t[k]))
        from evaluate_ate import ate
        out["ate"] = ate(a.out, gt_path, 0.02, True)
        est = np.loadtxt(a.out)[:, 1:4]
        out["unaligned_rmse_m"] = float(np.sqrt(np.mean(np.sum((est - gt_t[1:n]) ** 2, axis=1))))
    if a.save_volume:
        sdf.save(a.save_volume)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
