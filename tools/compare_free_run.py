#!/usr/bin/env python3
"""Free-running HIP tracker against the free-running CPU oracle from a common checkpoint.

The HIP path runs the reference's frame loop (sdf_reconstruction.cpp:69-74) up to frame START, the volume is
checkpointed (tsdf_save) and handed to the oracle together with the pose; then BOTH run on, each on its own state,
over the next FRAMES frames of the same images.  Nothing is teacher-forced after the hand-over: differences in the
last bits of A, b (other summation order) feed back through pose and volume, so the distance between the two
trajectories is the thing to watch.  Prints one JSON line; tests/test_gpu_sequence.py asserts on it.

  python3 tools/compare_free_run.py --voxels 128 --width 320 --height 240 --start 470 --frames 24
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compare(m=128, width=320, height=240, start=470, frames=24, noise=True, device="cuda", threads=None):
    import torch
    import oracle as orc
    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth
    threads = threads or max(1, min(16, os.cpu_count() or 1))
    seq = synth.Sequence(n_frames=start + frames + 1, width=width, height=height, noise=noise, holes=0.02 if noise else 0.0)
    s = ts.SDF(m)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    errors = 0
    for k in range(start + 1):                                   # the HIP path alone up to the hand-over
        xyz, nrm, rgb = (a.cpu().numpy() for a in seq.frame_torch(k, device))
        if k > 0:
            try:
                t.estimate_new_position(s, xyz)
            except ts.TsdfError:
                errors += 1
        s.update(t, xyz, nrm, rgb)
    with tempfile.TemporaryDirectory() as d:                     # through the checkpoint file, as a resumed run would
        path = os.path.join(d, "handover.tsdf")
        s.save(path)
        s2 = ts.SDF(m)
        s2.load(path)
    D, W = s2.download()
    cw, r, g, b = s2.download_color()
    s2.close()
    oo = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025)
    ot = orc.CameraTracking(oo)
    ot.set_K(seq.K)
    oo.D[:], oo.W[:], oo.Color_W[:], oo.R[:], oo.G[:], oo.B[:] = D, W, cw, r, g, b
    ot.set_camera_transformation(t.rot, t.trans)
    gap, it_g, it_o = [], [], []
    for k in range(start + 1, start + 1 + frames):
        xyz, nrm, rgb = (a.cpu().numpy() for a in seq.frame_torch(k, device))
        sg = t.estimate_new_position(s, xyz)
        so = ot.estimate_new_position(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
        s.update(t, xyz, nrm, rgb)
        oo.update(ot, orc.Cloud(xyz, nrm, rgb), threads=threads)
        gap.append(float(np.linalg.norm(t.trans - ot.trans)))
        it_g.append(int(sg["iterations"]))
        it_o.append(int(so["iterations"]))
    err_g = float(np.linalg.norm(t.trans - seq.t[start + frames]))
    return {"m": m, "image": [width, height], "start": start, "frames": frames, "track_errors_before_handover": errors,
            "max_gap_m": max(gap), "gap_m": gap, "iterations_hip": it_g, "iterations_oracle": it_o,
            "hip_error_vs_ground_truth_at_end_m": err_g, "path_length_m": float(np.sum(np.linalg.norm(np.diff(seq.t[start:start + frames + 1], axis=0), axis=1)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", dest="m", type=int, default=128)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--start", type=int, default=470)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--no-noise", action="store_true")
    a = ap.parse_args()
    print(json.dumps(compare(a.m, a.width, a.height, a.start, a.frames, not a.no_noise)))


if __name__ == "__main__":
    main()
