#!/usr/bin/env python3
"""Kernel-level micro-benchmark of the two hot kernels (for tuning and rocprofv3 --pmc passes).

Fusion-only mode (the reference's _useGroundTruth switch): frames are integrated at the true poses,
so every launch does the same work from run to run; then `--passes` tracker accumulation passes are
timed at a slightly wrong pose.  Prints one JSON line.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--passes", type=int, default=40)
    ap.add_argument("--no-color", action="store_true")
    ap.add_argument("--frame-step", type=int, default=8)
    ap.add_argument("--no-track-timing", action="store_true", help="time tracker passes by wall clock only (no events)")
    ap.add_argument("--mesh", type=int, default=0, help="also time N mesh extractions of the integrated volume")
    ap.add_argument("--look-along-k", action="store_true",
                    help="experiment: re-pose the camera so that it looks along the volume's fastest axis (k)")
    ap.add_argument("--roll", type=float, default=0.0, help="extra camera roll in degrees (exercises the row-major records)")
    args = ap.parse_args()

    import torch
    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth

    dev = torch.device("cuda", 0)
    seq = synth.Sequence(n_frames=args.frames, width=args.width, height=args.height, noise=True, holes=0.02,
                         step=args.frame_step)
    if args.roll:
        a = np.deg2rad(args.roll)
        Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        seq.R = np.array([R @ Rz for R in seq.R])
    if args.look_along_k:
        # the images stay what they are (camera frame); only the world pose changes: world -y (the viewing direction
        # of the reference's initial pose) is mapped to world +z, the camera moved to the bottom of the volume
        Q = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, -1.0, 0]])
        c = np.array([0.0, 1.0, -0.3])
        seq.t = np.array([Q @ t + c for t in seq.t])
        seq.R = np.array([Q @ R for R in seq.R])
    # frames are rendered at the ORIGINAL poses (camera-frame images do not change with --look-along-k), on the GPU
    render_seq = synth.Sequence(n_frames=args.frames, width=args.width, height=args.height, noise=True, holes=0.02,
                                step=args.frame_step)
    if args.roll:
        render_seq.R = np.array([R @ Rz for R in render_seq.R])
    d = [render_seq.frame_torch(k, dev) for k in range(args.frames)]
    torch.cuda.synchronize()
    sdf = ts.SDF(args.m, with_color=not args.no_color)
    trk = ts.CameraTracking(sdf=sdf)
    trk.set_K(seq.K)
    sdf.set_timing(True)
    per = []
    for rep in range(2):                       # second sweep = warm numbers
        sdf.read_timing(reset=True)
        sdf.read_counters(reset=True)
        for k in range(args.frames):
            trk.set_camera_transformation(seq.R[k], seq.t[k])
            sdf.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), args.width, args.height)
            sdf.update(want_stats=False)
        tm, cn = sdf.read_timing(), sdf.read_counters()
        per.append((tm, cn))
    tm, cn = per[-1]
    bpv = 16 if args.no_color else 48
    ms = tm["integrate_ms"] / tm["integrate_launches"]
    upd = cn["n_updated"] / tm["integrate_launches"]
    out = {"m": args.m, "integrate_ms": ms, "updated_per_launch": upd,
           "integrate_GBs": (bpv * upd + args.width * args.height * 27) / (ms * 1e-3) / 1e9,
           "pack_ms": tm["pack_ms"] / max(1, tm["pack_launches"]),
           "items_per_launch": cn["integrate_items"] / tm["integrate_launches"]}
    # tracker passes at a perturbed pose (not applied: accumulate only)
    k = args.frames - 1
    trk.set_camera_transformation(seq.R[k], seq.t[k] + np.array([0.01, -0.01, 0.005]))
    sdf.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), args.width, args.height)
    trk.accumulate()
    if args.no_track_timing:
        sdf.set_timing(False)
    sdf.read_timing(reset=True)
    sdf.read_counters(reset=True)
    import time
    t0 = time.perf_counter()
    for _ in range(args.passes):
        A, b, st = trk.accumulate()
    wall = (time.perf_counter() - t0) / args.passes
    tm, cn = sdf.read_timing(), sdf.read_counters()
    kms = tm["track_ms"] / tm["track_launches"] if tm["track_launches"] else None
    out.update({"track_pass_kernel_ms": kms, "track_pass_wall_ms": wall * 1e3,
                "track_in_grid": st["n_in_grid_owned"], "track_ok": st["n_ok"],
                "track_gather_GBs": 832.0 * st["n_in_grid_owned"] / (kms * 1e-3) / 1e9 if kms else None})
    if args.mesh:
        for color in (False, True) if not args.no_color else (False,):
            n = sdf.mesh(with_color=color, read=False)          # warm-up: buffers get allocated here
            t0 = time.perf_counter()
            for _ in range(args.mesh):
                n = sdf.mesh(with_color=color, read=False)
            dt = (time.perf_counter() - t0) / args.mesh
            out["mesh_color_ms" if color else "mesh_ms"] = dt * 1e3
            out["mesh_triangles"] = n
        out["mesh_sweep_GBs"] = 8.0 * args.m ** 3 / (out["mesh_ms"] * 1e-3) / 1e9      # D,W once per voxel
    print(json.dumps(out))


if __name__ == "__main__":
    main()
