"""SURVEY.md section 8d, config 1 ("plumbing"): the synthetic fr1/plant stream, m = 128, 10 frames, CPU restatement
only -- no GPU.  Checks the driver's sequencing (sdf_reconstruction.cpp:69-74), the pose-file format (:4-17: one
`timestamp tx ty tz qx qy qz qw` line per tracked frame, 4 decimals, appended), the timestamp association and the
ATE evaluation on the file pair."""
import os
import sys

import numpy as np

import oracle as orc
from tracking_sdf_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from evaluate_ate import ate, read_tum          # noqa: E402
from run_sequence import quat_from_rot          # noqa: E402


def test_ten_frames_at_128_on_the_cpu_restatement(tmp_path):
    n, m = 10, 128
    seq = synth.Sequence(n_frames=n, width=640, height=480, noise=True, holes=0.02)
    s = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025)          # sdf_reconstruction.cpp:83-85
    t = orc.CameraTracking(s, 20, 0.001, 1.0, 0.01)                        # :88
    t.set_K(seq.K)
    traj, gt = str(tmp_path / "trajectory.txt"), str(tmp_path / "groundtruth.txt")
    open(traj, "w").close()
    iters = []
    for frame_num in range(1, n + 1):
        xyz, nrm, rgb = seq.frame(frame_num - 1)
        cloud = orc.Cloud(xyz, nrm, rgb)
        if frame_num > 1:                                                   # :69-72
            st = t.estimate_new_position(s, cloud, threads=0, stale_carry=True)
            iters.append(st["iterations"])
            assert not st["nonfinite"]
            with open(traj, "a") as f:                                      # the reference appends, :10-16
                f.write("%.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n" % (seq.stamps[frame_num - 1], *t.trans, *quat_from_rot(t.rot)))
        assert s.update(t, cloud, with_color=True, threads=0) > 0           # :74
    with open(gt, "w") as f:
        f.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for k in range(n):
            f.write("%.4f %.6f %.6f %.6f 0 0 0 1\n" % (seq.stamps[k], *seq.t[k]))
    # file format: n - 1 lines (frame 1 is fused, not tracked), 8 columns, 4 decimals
    lines = open(traj).read().splitlines()
    assert len(lines) == n - 1
    for ln in lines:
        cols = ln.split()
        assert len(cols) == 8 and all(len(c.split(".")[1]) == 4 for c in cols)
    stamps, est = read_tum(traj)
    assert np.allclose(stamps, np.round(seq.stamps[1:n], 4))
    res = ate(traj, gt, 0.02, True)
    assert res["pairs"] == n - 1
    assert res["ate_rmse_m"] < 0.02                  # the tracker follows the path at a coarse 4.7 cm voxel size
    assert 1 <= min(iters) and max(iters) <= 20
    # the pose actually moved with the camera (not stuck at the initial pose)
    assert np.linalg.norm(est[-1] - est[0]) > 0.5 * np.linalg.norm(seq.t[n - 1] - seq.t[1])
