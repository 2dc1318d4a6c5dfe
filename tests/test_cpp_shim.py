"""The C++ shim (include/sdf_3d_reconstruction/hotpath.hpp) driven like the reference's frame loop.

CPU: the demo compiles with plain g++ against the C ABI (no HIP headers, no Eigen/PCL).
GPU: it runs the reference's sequencing (frame 1 integrate only, then track -> write pose -> integrate,
sdf_reconstruction.cpp:69-74) and its trajectory file matches the oracle's poses.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build_demo():
    subprocess.check_call(["make", "-C", ROOT, "-s", "shim_demo"])
    return os.path.join(ROOT, "build", "shim_demo")


def test_shim_demo_compiles_with_plain_gxx():
    exe = build_demo()
    assert os.access(exe, os.X_OK)
    # header must be self-contained C++17 without HIP / Eigen / PCL
    src = "#include \"sdf_3d_reconstruction/hotpath.hpp\"\nint main(){ tsdf_config c; tsdf_default_config(&c); return c.m == 256 ? 0 : 1; }\n"
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                       input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


def test_reference_call_sites_compile_unchanged_against_the_shim():
    """sdf_reconstruction.cpp:65,70-71,74,83-88,90-91 -- the reference's own statements with the reference's own
    types (Eigen::Vector3d&, pcl::PointCloud<...>::Ptr, Eigen-typed public trans / rot, camera_info_cb) -- compile
    against hotpath.hpp with -DTSDF_WITH_EIGEN_PCL -DTSDF_WITH_ROS.  Eigen / PCL / ROS are not in the image:
    tests/mock holds minimal stand-ins of the declarations touched (test infrastructure)."""
    src = os.path.join(ROOT, "tests", "mock", "callsite_compile.cpp")
    for std in ("c++11", "c++17"):          # the reference builds with -std=c++0x
        p = subprocess.run(["g++", "-std=" + std, "-fsyntax-only", "-Wall", "-Wextra", "-DTSDF_WITH_EIGEN_PCL", "-DTSDF_WITH_ROS",
                            "-I", os.path.join(ROOT, "tests", "mock"), "-I", os.path.join(ROOT, "include"), src],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
    # and links: header-only on top of the C ABI
    subprocess.check_call(["make", "-C", ROOT, "-s", "refcall_demo"])
    assert os.access(os.path.join(ROOT, "build", "refcall_demo"), os.X_OK)
    # without the macro nothing of Eigen / PCL is needed
    p = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                       input='#include "sdf_3d_reconstruction/hotpath.hpp"\nint main(){return 0;}\n', text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


@pytest.mark.gpu
def test_exact_type_call_sites_give_the_same_trajectory_as_the_plain_shim(tmp_path):
    from dump_frames import dump
    exe_plain = build_demo()
    subprocess.check_call(["make", "-C", ROOT, "-s", "refcall_demo"])
    exe_ref = os.path.join(ROOT, "build", "refcall_demo")
    frames_bin = str(tmp_path / "frames.bin")
    dump(frames_bin, n=4, width=160, height=120, step=2)
    outs = []
    # (the third run: the shim's three-argument estimate_new_position(sdf, cloud, normals) -- the whole frame staged under
    # the passes, update() integrates what was staged -- must give the same trajectory again)
    for exe, name, extra in ((exe_plain, "a.txt", []), (exe_ref, "b.txt", []), (exe_ref, "c.txt", ["normals-at-track"])):
        p = subprocess.run([exe, frames_bin, "64", str(tmp_path / name)] + extra, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        outs.append((np.loadtxt(str(tmp_path / name))[:, :4], [float(x) for x in p.stdout.split("=")[1].split()[:3]]))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][0], outs[2][0]) and outs[0][1] == outs[2][1]


@pytest.mark.gpu
def test_exact_type_shim_field_writes_constructor_constants_and_cloud_token(tmp_path):
    """Traps beyond sdf_reconstruction.cpp's own statements (camera_tracking.h:43-63): assigning the public pose / K fields,
    a CameraTracking built with other constants than the SDF's, a cloud modified in place between the two hot calls; and
    the rest of the two classes' public surface (sdf.h:113-181, camera_tracking.h:69-101) against the oracle: index and
    coordinate maps, projections (bit for bit) and get_partial_derivative (13 look-ups in HBM; J, value and the flag)."""
    import oracle as orc
    from dump_frames import dump
    subprocess.check_call(["make", "-C", ROOT, "-s", "shim_fields_demo"])
    frames_bin = str(tmp_path / "frames.bin")
    m = 64
    seq = dump(frames_bin, n=2, width=160, height=120, step=2)
    p = subprocess.run([os.path.join(ROOT, "build", "shim_fields_demo"), frames_bin, str(m)], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("ok ") == 4 and "FAIL" not in p.stdout
    # the helper values against the oracle's restatement of the same reference lines
    H = {}
    for line in p.stdout.splitlines():
        if line.startswith("H "):
            H.setdefault(line.split()[1], []).append([float(x) for x in line.split()[2:]])
    oo = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025)
    ot = orc.CameraTracking(oo)
    ot.set_K(seq.K)
    xyz, nrm, rgb = seq.frame(0)
    oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    L, C = orc.lib(), orc.C
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ijk = np.array([3, m - 1, 7], dtype=np.int32)
    g, vox = np.zeros(3), np.zeros(3)
    L.orc_get_global_coordinates(oo._p, ijk.ctypes.data_as(C.POINTER(C.c_int32)), dp(g))
    L.orc_get_voxel_coordinates(oo._p, dp(g), dp(vox))
    geo = H["geo"][0]
    assert geo[0:3] == list(g) and geo[3:6] == list(vox)
    cam = [(ot.rot_inv[r, 0] * g[0] + ot.rot_inv[r, 1] * g[1]) + ot.rot_inv[r, 2] * g[2] + ot.rot_inv_trans[r] for r in range(3)]
    assert geo[6:9] == cam
    K = np.asarray(seq.K, dtype=np.float64)
    ij = [(K[r, 0] * cam[0] + K[r, 1] * cam[1]) + K[r, 2] * cam[2] for r in range(3)]
    assert geo[9:11] == [ij[0] / ij[2], ij[1] / ij[2]]
    world = [(ot.rot[r, 0] * cam[0] + ot.rot[r, 1] * cam[1]) + ot.rot[r, 2] * cam[2] + ot.trans[r] for r in range(3)]
    assert geo[11:14] == world and np.allclose(world, g, atol=1e-12)
    assert H["idx"][0] == [m * m * 3 + m * (m - 1) + 7, -1, 3, m - 1, 7]
    n_ok = 0
    for row in H["J"]:
        cp = np.array(row[0:3], dtype=np.float32).astype(np.float64)
        in_grid, J, ok, val = ot.get_partial_derivative(oo, cp)
        assert in_grid                                     # (the demo's points lie inside the default volume)
        assert int(row[3]) == int(ok)
        assert row[4] == val or (np.isnan(row[4]) and np.isnan(val))
        if ok:
            assert row[5:11] == list(J)
            n_ok += 1
    assert n_ok > 10


@pytest.mark.gpu
def test_shim_demo_frame_loop_matches_oracle(tmp_path):
    import oracle as orc
    from dump_frames import dump
    exe = build_demo()
    frames_bin = str(tmp_path / "frames.bin")
    traj = str(tmp_path / "trajectory.txt")
    n, w, h, m = 5, 160, 120, 64
    seq = dump(frames_bin, n=n, width=w, height=h, step=2)
    p = subprocess.run([exe, frames_bin, str(m), traj], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = np.loadtxt(traj)
    assert got.shape == (n - 1, 8)
    # the same loop on the oracle
    oo = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025)
    ot = orc.CameraTracking(oo)
    ot.set_K(seq.K)
    want = []
    for k in range(n):
        xyz, nrm, rgb = seq.frame(k)
        cloud = orc.Cloud(xyz, nrm, rgb)
        if k > 0:
            ot.estimate_new_position(oo, cloud, threads=1, stale_carry=True)
            want.append(ot.trans.copy())
        oo.update(ot, cloud)
    want = np.array(want)
    assert np.allclose(got[:, 0], seq.stamps[1:], atol=1e-4)
    assert np.max(np.abs(got[:, 1:4] - want)) <= 5.1e-5          # file has 4 decimals
    final = [float(x) for x in p.stdout.strip().split("=")[1].split()]
    assert np.max(np.abs(np.array(final) - want[-1])) < 1e-6
    # the quaternion columns come from a det = -1 matrix (camera_tracking.cpp:7): only finiteness is meaningful
    assert np.all(np.isfinite(got[:, 4:8]))


def write_tum_dir(root, n, w, h, step=3):
    """A miniature TUM RGB-D directory (depth.txt + 16-bit depth PNGs, value = metres * 5000)."""
    from PIL import Image
    from tracking_sdf_amd import synth
    seq = synth.Sequence(n_frames=n, width=w, height=h, noise=True, holes=0.02, step=step)
    os.makedirs(os.path.join(root, "depth"), exist_ok=True)
    depths = []
    with open(os.path.join(root, "depth.txt"), "w") as f:
        f.write("# depth maps\n# timestamp filename\n")
        for k in range(n):
            z = seq.frame(k)[0][..., 2]
            d16 = np.where(np.isnan(z), 0, np.round(z * 5000.0)).astype(np.uint16)
            name = "depth/%.6f.png" % seq.stamps[k]
            Image.fromarray(d16).save(os.path.join(root, name))          # mode I;16
            f.write("%.6f %s\n" % (seq.stamps[k], name))
            depths.append(d16)
    return seq, depths


def test_offline_driver_compiles():
    subprocess.check_call(["make", "-C", ROOT, "-s", "sdf_offline"])
    assert os.access(os.path.join(ROOT, "build", "sdf_offline"), os.X_OK)


@pytest.mark.gpu
def test_offline_tum_driver_matches_python_binding(tmp_path):
    """tools/sdf_offline.cpp (C++: PNG decoding, GPU pre-processing, track, integrate) against the same loop through
    the Python binding on the same depth images: identical trajectories."""
    import tracking_sdf_amd as ts
    subprocess.check_call(["make", "-C", ROOT, "-s", "sdf_offline"])
    exe = os.path.join(ROOT, "build", "sdf_offline")
    root = str(tmp_path / "tum")
    n, w, h, m, rad = 5, 160, 120, 64, 6
    seq, depths = write_tum_dir(root, n, w, h)
    K = seq.K
    traj = str(tmp_path / "traj.txt")
    ply = str(tmp_path / "mesh.ply")
    p = subprocess.run([exe, root, str(m), traj, "0", str(K[0, 0]), str(K[1, 1]), str(K[0, 2]), str(K[1, 2]), str(rad), ply],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = np.loadtxt(traj)
    assert got.shape == (n - 1, 8)
    s = ts.SDF(m, with_color=False)
    t = ts.CameraTracking(sdf=s)
    t.set_K(K)
    want = []
    for k in range(n):
        s.set_depth_frame(depths[k], None, radius=rad)
        if k > 0:
            t.estimate_new_position()
            want.append(t.trans.copy())
        s.update()
    want = np.array(want)
    assert np.max(np.abs(got[:, 1:4] - want)) <= 5.1e-5                  # 4 decimals in the pose file
    assert np.max(np.abs(want - seq.t[1:n])) < 0.08                      # and it follows the true path (9.4 cm voxels, 160x120)
    assert '"track_errors": 0' in p.stdout
    # the mesh the driver wrote = the binding's mesh of the same volume, moved to the world frame
    v = s.mesh()
    assert len(v) > 100 and '"mesh_triangles": %d' % len(v) in p.stdout
    raw = open(ply, "rb").read()
    body = raw[raw.index(b"end_header\n") + len(b"end_header\n"):]
    got_v = np.frombuffer(body[:len(v) * 36], dtype="<f4").reshape(-1, 3, 3)
    want_v = (v.astype(np.float64) + np.array([-3.0, -3.0, -0.5])).astype(np.float32)
    assert np.array_equal(got_v, want_v)
    assert len(body) == len(v) * (36 + 13)


@pytest.mark.gpu
def test_offline_driver_fusion_only_mode(tmp_path):
    """The reference's _useGroundTruth switch (sdf_reconstruction.cpp:51-66) in the C++ driver: no tracking, every
    depth frame fused at the ground-truth pose nearest in time.  Poses come from a TUM-format file (quaternions), so
    the synthetic path is mirrored into a proper-rotation world first (x -> -x; the default volume is symmetric)."""
    import tracking_sdf_amd as ts
    from scipy.spatial.transform import Rotation
    subprocess.check_call(["make", "-C", ROOT, "-s", "sdf_offline"])
    exe = os.path.join(ROOT, "build", "sdf_offline")
    root = str(tmp_path / "tum")
    n, w, h, m, rad = 6, 160, 120, 64, 4
    seq, depths = write_tum_dir(root, n, w, h)
    M = np.diag([-1.0, 1.0, 1.0])
    Rs = [M @ seq.R[k] for k in range(n)]
    ts_ = [M @ seq.t[k] for k in range(n)]
    assert all(np.linalg.det(R) > 0.999 for R in Rs)
    gt = str(tmp_path / "groundtruth.txt")
    with open(gt, "w") as f:
        f.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for k in range(n):
            if k == 3:
                continue                                   # frame 3 has no pose within 20 ms: must be skipped
            q = Rotation.from_matrix(Rs[k]).as_quat()      # x y z w
            f.write("%.6f %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n" % (seq.stamps[k] + 0.004, *ts_[k], *q))
    K = seq.K
    traj, ply = str(tmp_path / "traj.txt"), str(tmp_path / "mesh.ply")
    p = subprocess.run([exe, root, str(m), traj, "0", str(K[0, 0]), str(K[1, 1]), str(K[0, 2]), str(K[1, 2]), str(rad), ply, gt],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert '"fusion_only": true' in p.stdout and '"frames_without_pose": 1' in p.stdout and '"frames": 5' in p.stdout
    assert os.path.getsize(traj) == 0                      # no tracking, no pose lines
    # the same loop through the Python binding
    s = ts.SDF(m, with_color=False)
    t = ts.CameraTracking(sdf=s)
    t.set_K(K)
    for k in range(n):
        if k == 3:
            continue
        s.set_depth_frame(depths[k], None, radius=rad)
        q = Rotation.from_matrix(Rs[k]).as_quat()
        t.set_camera_transformation(Rotation.from_quat(q).as_matrix(), ts_[k])
        s.update()
    v = s.mesh()
    n_cpp = int(p.stdout.split('"mesh_triangles": ')[1].split(",")[0])
    assert len(v) > 500 and abs(n_cpp - len(v)) <= 0.01 * len(v)      # poses agree to ~1e-16, not bit for bit
