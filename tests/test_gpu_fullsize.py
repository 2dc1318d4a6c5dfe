"""Full-size (BASELINE.json) configurations.

(1) HIP path against the CPU oracle AT the configurations' sizes (configs 2, 3, 4 and config 5's image size at the
largest volume whose oracle arrays are a reasonable ask of the host -- 1024^3, 26 GB): the regimes that exist only at size -- 8 chunks per voxel row, ~195 k work
items over 64 band regions + overflow, 1280 persistent workgroups with XCD feedback, 714 tracker workgroups over 8 shards,
640x480 / 1280x960 pixel-record layouts -- are checked bit for bit against sdf.cpp:224-315 / camera_tracking.cpp:66-363
as restated in oracle/tsdf_oracle.c (the oracle needs 61 ms per update and 11 ms per tracker call at 512^3 on the
box's 16 host threads).

(2) Size-independent properties where no oracle fits (2048^3: 206 GB of host arrays, and the reference's own int
voxel count wraps at m >= 1291, sdf.cpp:9).  Properties of SDF::update (sdf.cpp:289-292) that hold bit-exactly in float
arithmetic:
  * integrating the SAME frame at the SAME pose twice doubles W exactly and leaves D unchanged
    ((W d + w d')/(W + w) with identical terms is exact: x + x and 2x/2 never round);
  * the set of updated voxels is a function of pose and image only (counter identical both times);
  * a slab of the volume integrates to the same bits as the same layers of the whole volume.
Properties of the tracker: A is symmetric positive semi-definite, its term count equals the OK samples when
no sample is out of grid, and tracking a frame against the volume it was just fused into at the true pose
converges to (nearly) zero motion.
"""
import numpy as np
import pytest

from tracking_sdf_amd import synth
from util import assert_volume_equal_at_size, sym_rel_err

pytestmark = pytest.mark.gpu


def seq_frames(n, w=640, h=480, **kw):
    seq = synth.Sequence(n_frames=n, width=w, height=h, noise=True, holes=0.02, **kw)
    return seq, [seq.frame(k) for k in range(n)]


@pytest.mark.parametrize("m,color", [(512, True), (256, True)])
def test_double_integration_doubles_w_exactly(m, color):
    import tracking_sdf_amd as ts
    seq, fr = seq_frames(2, step=5)
    s = ts.SDF(m, with_color=color)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    xyz, nrm, rgb = fr[0]
    st1 = s.update(t, xyz, nrm, rgb)
    D1, W1 = s.download()
    st2 = s.update(t, xyz, nrm, rgb)
    D2, W2 = s.download()
    assert st1["n_updated"] == st2["n_updated"] and st1["n_voxels"] == m ** 3
    frac = st1["n_updated"] / m ** 3
    assert 0.03 < frac < 0.2                       # the frustum of a 640x480 camera inside the 6x6x3.5 m volume
    upd = W1 > 0
    assert int(upd.sum()) == st1["n_updated"]
    assert np.array_equal(W2[upd], 2 * W1[upd]) and np.array_equal(D2[upd], D1[upd])
    assert np.all(W2[~upd] == 0) and np.all(D2[~upd] == np.float32(15.5))
    assert np.all(np.abs(D1[upd]) <= np.float32(0.3))          # truncation: -delta clamp, d > delta skipped
    if color:
        cw, r, g, b = s.download_color()
        assert np.all(cw[~upd] == 0) and np.all(r[~upd] == np.float32(0.4))
        assert np.all((r[upd] >= 0) & (r[upd] <= 255))
    # second frame at its own pose: counters move, nothing becomes NaN
    t.set_camera_transformation(seq.R[1], seq.t[1])
    st3 = s.update(t, *fr[1])
    D3, W3 = s.download()
    assert st3["n_updated"] > 0 and not np.isnan(D3).any() and np.all(W3 >= W2)
    s.close()


def test_tracker_properties_at_512():
    import tracking_sdf_amd as ts
    m = 512
    seq, fr = seq_frames(3, step=3)
    s = ts.SDF(m, with_color=False)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    for k in range(2):
        t.set_camera_transformation(seq.R[k], seq.t[k])
        s.update(t, fr[k][0], fr[k][1])
    # accumulate at the true pose of frame 1
    s.set_frame(fr[1][0])
    A, b, st = t.accumulate()
    assert st["n_samples"] == 214 * 160 and st["n_oog"] == 0 and st["n_terms"] == st["n_ok"] > 25000
    assert np.array_equal(A, A.T) and np.all(np.linalg.eigvalsh(A) > -1e-6 * np.abs(A).max())
    # repeated passes are bitwise reproducible (fixed-order reduction, no float atomics)
    for _ in range(40):                      # 714 workgroups race to be the last arriver of their shard: same bits every time
        A2, b2, _ = t.accumulate()
        assert np.array_equal(A, A2) and np.array_equal(b, b2)
    # tracking frame 1 from its true pose stays put (sub-centimetre), frame 2 from pose 1 moves toward pose 2
    st = t.estimate_new_position(s, fr[1][0])
    assert np.linalg.norm(t.trans - seq.t[1]) < 0.01
    err_before = np.linalg.norm(seq.t[1] - seq.t[2])
    t.set_camera_transformation(seq.R[1], seq.t[1])
    st = t.estimate_new_position(s, fr[2][0])
    assert 1 <= st["iterations"] <= 20
    assert np.linalg.norm(t.trans - seq.t[2]) < 0.5 * err_before
    s.close()


def test_slab_of_512_matches_whole_volume_layers():
    import tracking_sdf_amd as ts
    m = 512
    seq, fr = seq_frames(1)
    whole = ts.SDF(m, with_color=False)
    tw = ts.CameraTracking(sdf=whole)
    tw.set_K(seq.K)
    whole.update(tw, fr[0][0], fr[0][1])
    Dw, Ww = whole.download()
    whole.close()
    x0, x1 = ts.slab_range(m, 8, 3)
    halo = ts.halo_for(ts.default_config(m=m), 6.0)
    slab = ts.SDF(m, with_color=False, slab=(x0, x1), halo=halo)
    tsl = ts.CameraTracking(sdf=slab)
    tsl.set_K(seq.K)
    st = slab.update(tsl, fr[0][0], fr[0][1])
    Ds, Ws = slab.download()
    sl = slice(x0 * m * m, x1 * m * m)
    assert np.array_equal(Ds, Dw[sl]) and np.array_equal(Ws, Ww[sl])
    assert st["n_voxels"] == (x1 - x0 + 2 * halo) * m * m and st["n_updated_halo"] > 0
    slab.close()


def surface_samples(s, seq, xyz, m):
    """SDF values at the voxel positions of a grid of valid pixels of the fused frame, and which of those pixels lie on
    a smooth piece of surface (no depth step within 3 pixels: a silhouette pixel sees distances of two surfaces mixed)."""
    rows, cols = np.arange(0, xyz.shape[0], 97), np.arange(0, xyz.shape[1], 89)
    z = xyz[..., 2]
    pts, smooth = [], []
    for r in rows:
        for c in cols:
            if not np.isfinite(xyz[r, c, 0]):
                continue
            win = z[max(r - 3, 0):r + 4, max(c - 3, 0):c + 4]
            win = win[np.isfinite(win)]                      # (random holes are not silhouettes)
            pts.append(xyz[r, c])
            smooth.append(bool(np.ptp(win) < 0.05))
    pts = np.asarray(pts)
    world = pts.astype(np.float64) @ seq.R[0].T + seq.t[0]
    vox = (world - np.array([-3.0, -3.0, -0.5])) * (m / np.array([6.0, 6.0, 3.5])) - 0.5
    val, ok = s.interpolate_distance(vox)
    return val, ok, np.asarray(smooth)


def test_config5_shapes_1280x960_at_1024():
    """Config 5's image size against a 1024^3 volume WITH the colour lanes (8 GiB of D/W + 16 GiB of colour on one GPU):
    counters and samples only -- the volume never leaves the device."""
    import tracking_sdf_amd as ts
    m = 1024
    seq, fr = seq_frames(2, w=1280, h=960, step=3)
    s = ts.SDF(m, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    st1 = s.update(t, *fr[0])
    st2 = s.update(t, *fr[0])
    assert st1["n_updated"] == st2["n_updated"] and 0.03 < st1["n_updated"] / m ** 3 < 0.2
    # sample the SDF at the voxel positions of some valid pixels: |value| must be below one truncation distance
    val, ok, smooth = surface_samples(s, seq, fr[0][0], m)
    # surface points sit at the zero crossing (one frame fused): away from silhouettes well inside one voxel diagonal,
    # everywhere within the truncation distance
    assert ok.mean() > 0.95 and smooth.mean() > 0.5
    assert np.all(np.abs(val[ok & smooth]) < 0.03) and np.all(np.abs(val[ok]) <= 0.3)
    s.set_frame(fr[1][0])
    A, b, st = t.accumulate()
    assert st["n_samples"] == 427 * 320 and st["n_ok"] > 100000 and np.array_equal(A, A.T)
    s.close()


FR3_K = np.array([[535.4, 0.0, 320.1], [0.0, 539.2, 247.6], [0.0, 0.0, 1.0]])


def mem_available_gb():
    """Host memory this job may still take: MemAvailable of the machine, capped by what the cgroup leaves (a container's
    limit does not show in /proc/meminfo) and by 256 GB -- nothing in this file has any business above that."""
    avail = 0.0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    avail = int(line.split()[1]) / 2 ** 20
    except OSError:
        return 0.0
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/memory"):
        try:
            with open(base + "/memory.max") as f:
                lim = f.read().strip()
            with open(base + "/memory.current") as f:
                cur = int(f.read().strip())
            if lim != "max":
                avail = min(avail, (int(lim) - cur) / 2 ** 30)
        except (OSError, ValueError):
            pass
    return min(avail, 256.0)


# (config of BASELINE.json, m, image, intrinsics, host GB the oracle + the downloads need)
ORACLE_AT_SIZE = [
    pytest.param(256, 640, 480, None, 2, id="config2-256-640x480"),
    pytest.param(512, 640, 480, None, 12, id="config3-512-640x480"),
    pytest.param(1024, 640, 480, FR3_K, 60, id="config4-1024-640x480-fr3"),
    pytest.param(1024, 1280, 960, None, 60, id="config5-image-1280x960-at-1024"),
    # (config 5 itself, 2048^3, is NOT oracle-checked: 206 GB of oracle arrays next to 206 GB of downloaded volume took the
    # GPU box down when it was tried -- its 3 TB are the machine's, not the job's -- and the reference cannot run that size
    # anyway: its int voxel count wraps at m >= 1291, sdf.cpp:9.  The property tests below cover 2048^3.)
]
CARRY_THREADS = 4


@pytest.mark.parametrize("m,w,h,K,need_gb", ORACLE_AT_SIZE)
def test_hip_path_equals_the_oracle_at_baseline_size(m, w, h, K, need_gb):
    """SDF::update (sdf.cpp:224-315) and estimate_new_position (camera_tracking.cpp:66-245) at the sizes BASELINE.json
    names, HIP against the oracle on the same inputs: three noisy frames with 2 % holes fused at ground-truth poses (the
    third with the camera rolled by 35 degrees: row-major pixel records, other image bands), n_updated equal per frame,
    D / W / Color_W / R / G / B bit-exact (exp() band <= 1 ulp of W); then, on the oracle's volume, one
    Gauss-Newton pass (counts equal, A, b <= 1e-11) and one whole tracker call (same iterations and stop flag, pose
    <= 1e-9) at the reference's thread-local carry state for CARRY_THREADS OpenMP threads."""
    import oracle as orc
    import tracking_sdf_amd as ts
    if mem_available_gb() < need_gb:
        pytest.skip(f"{need_gb} GB of host memory needed for the oracle's {m}^3 arrays next to the downloaded volume "
                    f"({mem_available_gb():.0f} GB available): BASELINE config {m}^3 {w}x{h} not oracle-checked on this host")
    seq = synth.Sequence(n_frames=4, width=w, height=h, noise=True, holes=0.02, step=4, K=K)
    a = np.deg2rad(35.0)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    fused = [(seq.R[0], seq.t[0], seq.frame(0)), (seq.R[1], seq.t[1], seq.frame(1)),
             (seq.R[2] @ Rz, seq.t[2], synth.render_frame(seq.R[2] @ Rz, seq.t[2], seq.K, w, h, noise=True, holes=0.02,
                                                          rng=np.random.default_rng(5)))]
    oo = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025, with_global_coords=(m <= 512))
    oo.track_exp_band()              # m^3 bytes: which voxels took the exp() weight (the only ones allowed to differ, util.py)
    ot = orc.CameraTracking(oo)
    ot.set_K(seq.K)
    go = ts.SDF(m, with_color=True, carry_threads=CARRY_THREADS)
    gt = ts.CameraTracking(sdf=go)
    gt.set_K(seq.K)
    for R, t, (xyz, nrm, rgb) in fused:
        ot.set_camera_transformation(R, t)
        gt.set_camera_transformation(R, t)
        n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
        st = go.update(gt, xyz, nrm, rgb)
        assert st["n_updated"] == n_or and st["n_voxels"] == m ** 3 and n_or > 0.02 * m ** 3
    n_exp = assert_volume_equal_at_size(go, oo, color=True)
    # the tracker on IDENTICAL volumes (the <= 1-ulp exp() voxels would otherwise show up in A at 1e-9)
    if n_exp:
        go.upload(oo.D, oo.W)
    xyz = seq.frame(3)[0]
    cloud = orc.Cloud(xyz)
    ot.set_camera_transformation(seq.R[2], seq.t[2])
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    A_o, b_o, st_o = ot.accumulate(oo, cloud, threads=CARRY_THREADS, stale_carry=True)
    go.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    assert st_g["n_samples"] == st_o["n_samples"] == -(-w // 3) * -(-h // 3)
    for key in ("n_nan", "n_oog", "n_ok", "n_terms"):
        assert st_g[key] == st_o[key], key
    assert st_o["n_ok"] > 0.6 * st_o["n_samples"]
    assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11
    so = ot.estimate_new_position(oo, cloud, threads=CARRY_THREADS, stale_carry=True)
    sg = gt.estimate_new_position(go, xyz)
    assert sg["iterations"] == so["iterations"] and bool(sg["stopped"]) == so["stopped"] and not so["nonfinite"]
    assert sg["n_terms_last"] == so["n_terms_last"]
    assert np.max(np.abs(gt.trans - ot.trans)) < 1e-9 and np.max(np.abs(gt.rot - ot.rot)) < 1e-9
    assert np.linalg.norm(gt.trans - seq.t[3]) < np.linalg.norm(seq.t[2] - seq.t[3])       # and it moved the right way
    go.close()


def _seed_sweep():
    import os
    n = int(os.environ.get("TSDF_PARITY_SEEDS", "2"))
    m = int(os.environ.get("TSDF_PARITY_SEEDS_M", "256"))
    return [pytest.param(m, 101 + 17 * i, id=f"{m}-seed{101 + 17 * i}") for i in range(n)]


@pytest.mark.parametrize("m,seed", _seed_sweep())
def test_hip_path_equals_the_oracle_on_other_noise_seeds(m, seed):
    """The bars of test_hip_path_equals_the_oracle_at_baseline_size on OTHER noise / hole patterns and another roll of the
    camera per seed (ten seeds at size; the suite runs two at 256^3, TSDF_PARITY_SEEDS=10
    TSDF_PARITY_SEEDS_M=512 runs the ten at 512^3 -- recorded in profiles/r06_parity_seeds.json).  Three frames fused at
    ground-truth poses on both sides, all six arrays: differences only in voxels that took the exp() weight, there W <= 1 ulp,
    D and colour <= 4 ulp; then a whole tracker call on identical volumes: same iterations, stop flag, pose <= 1e-9."""
    import json, os
    import oracle as orc
    import tracking_sdf_amd as ts
    w, h = 640, 480
    rng = np.random.default_rng(seed)
    seq = synth.Sequence(n_frames=4, width=w, height=h, noise=True, holes=float(rng.uniform(0.0, 0.05)), step=int(rng.integers(2, 6)), seed=seed)
    a = np.deg2rad(float(rng.uniform(-60.0, 60.0)))
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    fused = [(seq.R[0], seq.t[0], seq.frame(0)), (seq.R[1], seq.t[1], seq.frame(1)),
             (seq.R[2] @ Rz, seq.t[2], synth.render_frame(seq.R[2] @ Rz, seq.t[2], seq.K, w, h, noise=True, holes=0.02,
                                                          rng=np.random.default_rng(seed + 1)))]
    oo = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025, with_global_coords=True)
    oo.track_exp_band()
    ot = orc.CameraTracking(oo)
    ot.set_K(seq.K)
    go = ts.SDF(m, with_color=True, carry_threads=CARRY_THREADS)
    gt = ts.CameraTracking(sdf=go)
    gt.set_K(seq.K)
    updated = []
    for R, t, (xyz, nrm, rgb) in fused:
        ot.set_camera_transformation(R, t)
        gt.set_camera_transformation(R, t)
        n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
        st = go.update(gt, xyz, nrm, rgb)
        assert st["n_updated"] == n_or and n_or > 0.02 * m ** 3
        updated.append(int(n_or))
    n_exp = assert_volume_equal_at_size(go, oo, color=True)
    if n_exp:
        go.upload(oo.D, oo.W)
    xyz = seq.frame(3)[0]
    ot.set_camera_transformation(seq.R[2], seq.t[2])
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    so = ot.estimate_new_position(oo, orc.Cloud(xyz), threads=CARRY_THREADS, stale_carry=True)
    sg = gt.estimate_new_position(go, xyz)
    assert sg["iterations"] == so["iterations"] and bool(sg["stopped"]) == so["stopped"] and not so["nonfinite"]
    assert sg["n_terms_last"] == so["n_terms_last"]
    dpose = float(max(np.max(np.abs(gt.trans - ot.trans)), np.max(np.abs(gt.rot - ot.rot))))
    assert dpose < 1e-9
    go.close()
    out = os.environ.get("TSDF_PARITY_SEEDS_OUT")
    if out:
        with open(out, "a") as f:
            f.write(json.dumps({"m": m, "seed": seed, "roll_deg": float(np.rad2deg(a)), "holes": seq.holes, "step": seq.step if hasattr(seq, "step") else None,
                                "n_updated": updated, "voxels_in_the_exp_band_that_differ": int(n_exp), "tracker_iterations": int(sg["iterations"]),
                                "max_pose_difference": dpose}) + "\n")


def test_config4_workload_1024_cubed_with_colour_and_fr3_intrinsics():
    """BASELINE config 4 as it is specified: fr3 intrinsics (535.4 / 539.2 / 320.1 / 247.6), 640x480 images, 1024^3
    voxels WITH the colour lanes the reference always updates (sdf.cpp:294-304): 8 GiB of {D,W} + 16 GiB of colour on
    the one GPU (the 8-way slab split of this volume is test_multirank_gpu / bench --config 4).  The volume never leaves
    the device: counters, device-side SDF samples, and the colours of the extracted mesh."""
    import tracking_sdf_amd as ts
    m = 1024
    seq = synth.Sequence(n_frames=2, width=640, height=480, noise=True, holes=0.02, step=3, K=FR3_K)
    fr = [seq.frame(k) for k in range(2)]
    s = ts.SDF(m, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    st1 = s.update(t, *fr[0])
    # colours after ONE update are the pixel's own bytes: Color_W = wc, R = (0 * 0.4 + wc * r) / wc = r (up to rounding)
    v1, c1 = s.mesh(with_color=True)
    st2 = s.update(t, *fr[0])
    assert st1["n_voxels"] == m ** 3 and st1["n_updated"] == st2["n_updated"] and 0.03 < st1["n_updated"] / m ** 3 < 0.2
    val, ok, smooth = surface_samples(s, seq, fr[0][0], m)
    assert ok.mean() > 0.95 and smooth.mean() > 0.5
    assert np.all(np.abs(val[ok & smooth]) < 0.03) and np.all(np.abs(val[ok]) <= 0.3)
    # the same frame twice leaves D where it was ((W d + w d) / (W + w) with identical terms is exact): the mesh does not
    # move; the colour averages move by roundings only ((wc r) / wc is r up to an ulp, then averaged with r again)
    v2, c2 = s.mesh(with_color=True)
    assert len(v1) > 200000 and np.array_equal(v1.view(np.int32), v2.view(np.int32))
    rgb1, rgb2 = c1[..., :3], c2[..., :3]
    fin = np.isfinite(rgb1).all(axis=-1) & np.isfinite(rgb2).all(axis=-1)
    assert fin.mean() > 0.99 and np.allclose(rgb1[fin], rgb2[fin], rtol=1e-5, atol=1e-4)
    assert np.nanmin(rgb1) >= 0.0 and np.nanmax(rgb1) <= 255.0 and np.nanmax(rgb1) > 0.5      # real colours, not the 0.4 grey of the constructor
    s.set_frame(fr[1][0])
    t.set_camera_transformation(seq.R[1], seq.t[1])
    A, b, st = t.accumulate()
    assert st["n_samples"] == 214 * 160 and st["n_ok"] > 25000 and np.array_equal(A, A.T)
    s.close()


def test_config5_2048_cubed_on_one_gpu():
    """BASELINE config 5 is 2048^3 over 8 GPUs; one MI355X (288 GB) holds the whole volume by itself -- 64 GiB of {D,W}
    AND the 128 GiB of colour the reference always updates (sdf.cpp:294-304) -- which exercises 64-bit voxel indexing
    (the reference's int voxel count wraps at m >= 1291, sdf.cpp:9) and the 4 M-row work list.  Counters and
    device-side samples only."""
    import tracking_sdf_amd as ts
    m = 2048
    seq, fr = seq_frames(2, w=1280, h=960, step=3)
    s = ts.SDF(m, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    st1 = s.update(t, *fr[0])
    st2 = s.update(t, *fr[0])
    assert st1["n_voxels"] == m ** 3 and st1["n_updated"] == st2["n_updated"]
    assert 0.03 < st1["n_updated"] / m ** 3 < 0.2
    val, ok, smooth = surface_samples(s, seq, fr[0][0], m)
    assert ok.mean() > 0.95 and smooth.mean() > 0.5
    assert np.all(np.abs(val[ok & smooth]) < 0.03) and np.all(np.abs(val[ok]) <= 0.3)
    # voxels at the far corner of the volume (linear index > 2^31) keep their constructor value
    far = np.array([[m - 1.0, m - 1.0, m - 1.0], [m - 2.0, m - 1.0, 5.0]])
    val, ok = s.interpolate_distance(far)
    assert not ok.any()
    s.set_frame(fr[1][0])
    t.set_camera_transformation(seq.R[1], seq.t[1])
    A, b, st = t.accumulate()
    assert st["n_ok"] > 100000 and np.array_equal(A, A.T)
    s.close()


def test_mesh_at_512_properties():
    """Mesh extraction at the headline volume size: the triangle count equals a NumPy count of the case numbers of
    the downloaded volume, every vertex sits on a cube edge, and extraction is repeatable bit for bit."""
    import os
    import sys
    import tracking_sdf_amd as ts
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_mc_tables as gen
    m = 512
    seq, fr = seq_frames(4, step=10)
    s = ts.SDF(m, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    for k in range(4):
        t.set_camera_transformation(seq.R[k], seq.t[k])
        s.update(t, *fr[k])
    v, c = s.mesh(with_color=True)
    assert len(v) > 100000 and c.shape == (len(v), 3, 4)
    v2 = s.mesh()
    assert np.array_equal(v.view(np.int32), v2.view(np.int32))
    # NumPy count, layer by layer (marching_cubes_sdf.cpp:108-115, :203-239)
    ntri = np.array([len(x) for x in gen.build()], dtype=np.int64)
    D, W = s.download()
    D = D.reshape(m, m, m); W = W.reshape(m, m, m)
    offs = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]
    total = 0
    for i in range(1, m - 1):
        idx = np.zeros((m - 2, m - 2), dtype=np.int64)
        ok = np.ones((m - 2, m - 2), dtype=bool)
        for cnr, (dx, dy, dz) in enumerate(offs):
            sl = (i + dx, slice(1 + dy, m - 1 + dy), slice(1 + dz, m - 1 + dz))
            idx |= (D[sl] < 0).astype(np.int64) << cnr
            ok &= W[sl] > 0
        total += int(ntri[idx[ok]].sum())
    assert total == len(v)
    # every vertex lies on an edge of the reference's cube lattice (spacing extent/m, no half-voxel offset)
    cell = np.array([6.0 / m, 6.0 / m, 3.5 / m])
    q = v.reshape(-1, 3).astype(np.float64) / cell
    off = np.abs(q - np.round(q))
    assert np.all(np.sort(off, axis=1)[:, 1] < 1e-3)
    # colours: interpolated values are /255-scaled, exact hits raw 0..255, never negative
    rgb = c[..., :3]
    assert np.all(c[..., 3] == 1.0) and np.nanmin(rgb) >= 0.0 and np.nanmax(rgb) <= 255.0
    s.close()
