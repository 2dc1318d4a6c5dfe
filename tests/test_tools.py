"""CPU tests of the offline tooling: ATE evaluator (alignment incl. the mirrored world) and pose-file format."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from evaluate_ate import ate  # noqa: E402


def write_tum(path, stamps, pos):
    with open(path, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for s, p in zip(stamps, pos):
            f.write("%.4f %.6f %.6f %.6f 0 0 0 1\n" % (s, *p))


def test_ate_alignment_handles_rigid_motion_and_reflection(tmp_path):
    rng = np.random.default_rng(0)
    stamps = 1305031000.0 + np.arange(200) / 30.0
    gt = np.cumsum(rng.standard_normal((200, 3)) * 0.01, axis=0)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    est = gt @ q.T + np.array([0.3, -1.0, 2.0])
    noise = rng.standard_normal(est.shape) * 0.004
    g, e = str(tmp_path / "gt.txt"), str(tmp_path / "est.txt")
    write_tum(g, stamps, gt)
    write_tum(e, stamps + 0.003, est + noise)
    r = ate(e, g)
    assert r["pairs"] == 200 and abs(r["ate_rmse_m"] - 0.004 * np.sqrt(3)) < 0.001 and r["alignment_det"] > 0
    # a mirrored world (the reference's det = -1 initial pose): only a reflection-tolerant alignment recovers it
    mirror = np.diag([1.0, -1.0, 1.0])
    write_tum(e, stamps, est @ mirror)
    r_ref = ate(e, g, allow_reflection=True)
    r_prop = ate(e, g, allow_reflection=False)
    assert r_ref["ate_rmse_m"] < 1e-5 and r_ref["alignment_det"] < 0
    assert r_prop["ate_rmse_m"] > 10 * r_ref["ate_rmse_m"] + 1e-3
    # association drops poses without a ground-truth neighbour
    write_tum(e, np.concatenate([stamps[:50], stamps[50:100] + 5000.0]), est[:100])
    assert ate(e, g)["pairs"] == 50


def test_quaternion_of_pose_file_matches_scipy_for_proper_rotations():
    from run_sequence import quat_from_rot
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    for _ in range(50):
        R = Rotation.random(random_state=rng.integers(1 << 30)).as_matrix()
        q = quat_from_rot(R)
        want = Rotation.from_matrix(R).as_quat()
        assert np.allclose(q, want, atol=1e-12) or np.allclose(q, -want, atol=1e-12)


def test_bench_launcher_reports_a_failed_rank_without_a_gpu():
    """bench.py --gpus 2 starts two rank processes itself; here (no GPU) both refuse to run, and the launcher must
    come back non-zero with no result line instead of pretending a one-rank run."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "no GPU visible" in p.stderr and "[bench] rank" in p.stderr
