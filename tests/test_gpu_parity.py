"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (stated per SURVEY.md section 8c):
  D, W, colour lanes   bit-exact, except voxels whose weight went through exp() (eps <= d <= delta):
                       glibc exp vs ROCm ocml exp may differ in the last f64 bit, which after the
                       f64->f32 narrowing is at most 1 float ulp on W (and what follows from it on D)
  interpolation        bit-exact
  A, b (6x6 / 6)       <= 1e-11 relative to the largest entry (f64, different summation order)
  pose, one GN call    <= 1e-9
  pose, 10 free frames <= 1e-5 m
"""
import os

import numpy as np
import pytest

import oracle as orc
from tracking_sdf_amd import synth
from util import VOL, assert_volume_equal_at_size, make_gpu, make_oracle, scaled_K, sym_rel_err, ulp_diff

pytestmark = pytest.mark.gpu

W_, H_ = 160, 120


def frames(n, noise=False, holes=0.0, width=W_, height=H_, step=4):
    seq = synth.Sequence(n_frames=n, width=width, height=height, noise=noise, holes=holes, step=step)
    return seq, [seq.frame(k) for k in range(n)]


def assert_volume_equal(go, oo, m, color=True, max_exp_ulp=1):
    """bit-exact, except voxels whose weight went through exp() (recorded by the oracle): W <= 1 ulp,
    D / colour <= 4 ulp there, < 1e-4 of the voxels in all (tests/util.py: the same bar at every size)."""
    n_bad = assert_volume_equal_at_size(go, oo, color=color, max_exp_ulp=max_exp_ulp)
    return n_bad / float(m ** 3)


@pytest.mark.parametrize("m", [32, 48, 64])
def test_integrate_one_frame(m):
    seq, fr = frames(1)
    K = seq.K
    oo, ot = make_oracle(m, K)
    go, gt = make_gpu(m, K)
    xyz, nrm, rgb = fr[0]
    n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb), with_color=True)
    st = go.update(gt, xyz, nrm, rgb)
    assert st["n_updated"] == n_or and st["n_voxels"] == m ** 3 and st["n_updated_halo"] == 0
    assert_volume_equal(go, oo, m)


def test_integrate_sequence_ground_truth_poses():
    """Fusion-only mode (the reference's _useGroundTruth switch): 6 frames at the true poses."""
    m = 64
    seq, fr = frames(6, noise=True, holes=0.02)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K)
    for k, (xyz, nrm, rgb) in enumerate(fr):
        ot.set_camera_transformation(seq.R[k], seq.t[k])
        gt.set_camera_transformation(seq.R[k], seq.t[k])
        n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
        st = go.update(gt, xyz, nrm, rgb)
        assert st["n_updated"] == n_or
    assert_volume_equal(go, oo, m)


def test_integrate_sequence_without_colour_lanes():
    """6 noisy frames with holes at the true poses with the colour lanes off (24-byte pixel records, D / W only)."""
    m = 64
    seq, fr = frames(6, noise=True, holes=0.02)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K, with_color=False)
    for k, (xyz, nrm, rgb) in enumerate(fr):
        ot.set_camera_transformation(seq.R[k], seq.t[k])
        gt.set_camera_transformation(seq.R[k], seq.t[k])
        n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb), with_color=False)
        st = go.update(gt, xyz, nrm, None)
        assert st["n_updated"] == n_or
    assert_volume_equal(go, oo, m, color=False)


def test_integrate_without_color_lanes():
    m = 32
    seq, fr = frames(1)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K, with_color=False)
    xyz, nrm, rgb = fr[0]
    oo.update(ot, orc.Cloud(xyz, nrm, rgb), with_color=False)
    go.update(gt, xyz, nrm)
    assert_volume_equal(go, oo, m, color=False)


def _fused_pair(m, n_fuse=3, noise=False, holes=0.0, **kw):
    seq, fr = frames(n_fuse + 1, noise=noise, holes=holes)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K, **kw)
    for k in range(n_fuse):
        xyz, nrm, rgb = fr[k]
        ot.set_camera_transformation(seq.R[k], seq.t[k])
        oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    go.upload(oo.D, oo.W)      # identical volumes on both sides
    return seq, fr, oo, ot, go, gt


def test_interpolate_distance_bit_exact():
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-2.0, m + 1.0, size=(20000, 3))
    pts[:2000] = np.round(pts[:2000])                      # exact corner hits
    pts[2000:3000, 1] = np.round(pts[2000:3000, 1])
    pts[3000] = [-0.5, -0.25, 0.75]
    pts[3001] = [1e12, 3.0, 3.0]
    pts[3002] = [-1e12, 3.0, 3.0]
    val, ok = go.interpolate_distance(pts)
    for i in range(len(pts)):
        v, k = oo.interpolate_distance(pts[i])
        assert bool(ok[i]) == k, i
        if k:
            assert np.float32(v) == val[i], (i, pts[i], v, val[i])
        else:
            assert np.isnan(val[i])


@pytest.mark.parametrize("stale", [True, False])
@pytest.mark.parametrize("holes", [0.0, 0.05])
def test_accumulate_matches_oracle(stale, holes):
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m, holes=holes, stale_carry=stale)
    k = 3
    xyz = fr[k][0]
    for pose in ((seq.R[k], seq.t[k]), (seq.R[k - 1], seq.t[k - 1])):
        ot.set_camera_transformation(*pose)
        gt.set_camera_transformation(*pose)
        A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=stale)
        go.set_frame(xyz)
        A_g, b_g, st_g = gt.accumulate()
        assert st_g["n_samples"] == st_o["n_samples"] and st_g["n_nan"] == st_o["n_nan"]
        assert st_g["n_oog"] == st_o["n_oog"] and st_g["n_ok"] == st_o["n_ok"]
        assert st_g["n_in_grid_owned"] == st_o["n_ok"] + st_o["n_fail"]
        assert st_g["n_terms"] == st_o["n_terms"] and st_o["n_ok"] > 500
        assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11
        assert np.array_equal(A_g, A_g.T)


def test_accumulate_stale_carry_with_out_of_grid_samples():
    """A volume smaller than the scene puts whole image regions outside the grid, so the reference's
    carry-over (camera_tracking.cpp:156-159,261-268) fires on long runs, across wavefront and
    workgroup boundaries of the tracker kernel."""
    m = 48
    vol = dict(width=2.4, height=2.4, depth=2.4, origin=(-1.2, -3.0, -0.2), delta=0.3, epsilon=0.025)
    seq, fr = frames(3, holes=0.03, width=320, height=240)
    oo, ot = make_oracle(m, seq.K, vol)
    go, gt = make_gpu(m, seq.K, vol)
    for k in range(2):
        xyz, nrm, rgb = fr[k]
        ot.set_camera_transformation(seq.R[k], seq.t[k])
        oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    go.upload(oo.D, oo.W)
    ot.set_camera_transformation(seq.R[2], seq.t[2])
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    xyz = fr[2][0]
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
    A_n, b_n, st_n = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=False)
    go.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    assert st_o["n_oog"] > 1000 and st_o["n_terms"] > st_n["n_terms"] + 20     # the carry really fires
    assert st_g["n_terms"] == st_o["n_terms"] and st_g["n_oog"] == st_o["n_oog"]
    assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11


def test_tracker_constants_set_after_creation_reach_the_kernel():
    """camera_tracking.cpp:3-17: the reference's constructor accepts any v_h / w_h and forms its finite-difference
    denominators from them.  A volume created with the defaults whose CameraTracking is then constructed with other steps
    (tsdf_set_tracker_params) must perturb by AND divide by the new steps: same A, b as a handle created with those
    steps, and as the oracle (round 3 divided by the old denominators)."""
    import tracking_sdf_amd as ts
    m = 64
    gn = (20, 0.001, 0.5, 0.02)
    seq, fr, oo, _, go_ref, gt_ref = _fused_pair(m, gn=gn)
    ot = orc.CameraTracking(oo, *gn)
    ot.set_K(seq.K)
    go = ts.SDF(m, VOL["width"], VOL["height"], VOL["depth"], VOL["origin"], VOL["delta"], VOL["epsilon"])   # default steps
    gt = ts.CameraTracking(*gn, go)                         # ... replaced here
    gt.set_K(seq.K)
    go.upload(oo.D, oo.W)
    k = 3
    xyz = fr[k][0]
    for t_ in (ot, gt, gt_ref):
        t_.set_camera_transformation(seq.R[k - 1], seq.t[k - 1])
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
    go.set_frame(xyz); go_ref.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    A_r, b_r, st_r = gt_ref.accumulate()
    assert st_g["n_terms"] == st_o["n_terms"] == st_r["n_terms"] and st_o["n_ok"] > 500
    assert np.array_equal(A_g, A_r) and np.array_equal(b_g, b_r)
    assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11
    # and the default steps give something else: the test would notice a kernel that ignored the new ones
    _, _, _, _, go_d, gt_d = _fused_pair(m)
    gt_d.set_camera_transformation(seq.R[k - 1], seq.t[k - 1])
    go_d.set_frame(xyz)
    A_d, _, _ = gt_d.accumulate()
    assert sym_rel_err(A_d, A_o) > 1e-3


CARRY_THREADS = [1, 2, 3, 8, 16]


@pytest.mark.parametrize("threads", CARRY_THREADS)
def test_accumulate_reproduces_the_reference_at_its_thread_count(threads):
    """camera_tracking.cpp:72-76,146-162: the carry state is thread-local and starts fresh in every OpenMP thread's column
    chunk, so the real binary on an n-core host produces the np = n sums.  tsdf_config.carry_threads reproduces them
    (oracle: real OpenMP threads, static schedule): identical counts, A and b to 1e-11."""
    m = 48
    vol = dict(width=2.4, height=2.4, depth=2.4, origin=(-1.2, -3.0, -0.2), delta=0.3, epsilon=0.025)
    seq, fr = frames(3, holes=0.03, width=320, height=240)
    oo, ot = make_oracle(m, seq.K, vol)
    go, gt = make_gpu(m, seq.K, vol, carry_threads=threads)
    for k in range(2):
        xyz, nrm, rgb = fr[k]
        ot.set_camera_transformation(seq.R[k], seq.t[k])
        oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    go.upload(oo.D, oo.W)
    ot.set_camera_transformation(seq.R[2], seq.t[2])
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    xyz = fr[2][0]
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=threads, stale_carry=True)
    A_1, b_1, st_1 = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
    go.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    assert st_o["n_terms"] <= st_1["n_terms"]
    if threads >= 8:
        assert st_o["n_terms"] < st_1["n_terms"]            # chunk starts really cut runs on this frame
    assert st_g["n_terms"] == st_o["n_terms"] and st_g["n_oog"] == st_o["n_oog"] and st_g["n_ok"] == st_o["n_ok"]
    assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11


@pytest.mark.parametrize("threads", [2, 3, 7, 16, 64])
def test_long_out_of_grid_run_is_cut_at_chunk_starts(threads):
    """The run of test_long_out_of_grid_run_spans_workgroups (one valid sample, 900 out-of-grid ones over 30 columns,
    a valid one) under np threads: the run ends with the first thread's columns; a sample behind a chunk start without a
    valid predecessor in its own chunk adds nothing."""
    m = 32
    vol = dict(width=3.2, height=3.2, depth=3.2, origin=(0.0, 0.0, 0.0), delta=0.3, epsilon=0.025)
    K = scaled_K(64, 48)
    oo, ot = make_oracle(m, K, vol)
    oo.create_circle(1.0, 1.6, 1.6, 1.6)
    go, gt = make_gpu(m, K, vol, carry_threads=threads)
    go.upload(oo.D, oo.W)
    eye, zero = np.eye(3), np.zeros(3)
    ot.set_camera_transformation(eye, zero)
    gt.set_camera_transformation(eye, zero)
    ncol, nrow = 40, 30
    xyz = np.full((3 * nrow - 2, 3 * ncol - 2, 3), np.nan, dtype=np.float32)
    samp = np.full((ncol, nrow, 3), [-5.0, 1.0, 1.0], dtype=np.float32)      # all out of grid
    samp[0, 5] = (1.7, 1.5, 1.4)
    samp[0, 7] = (np.nan, 0, 0)
    samp[13, 29] = (1.5, 1.6, 1.7)             # last sample of a column: its run starts in the next column
    samp[30, 2] = (1.2, 1.9, 1.6)
    for c in range(ncol):
        for r in range(nrow):
            xyz[3 * r, 3 * c] = samp[c, r]
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=threads, stale_carry=True)
    go.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    assert st_g["n_terms"] == st_o["n_terms"] and st_g["n_ok"] == st_o["n_ok"]
    assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11


def test_long_out_of_grid_run_spans_workgroups():
    """One valid sample followed by > 3 workgroups' worth of out-of-grid samples, then a valid one."""
    m = 32
    vol = dict(width=3.2, height=3.2, depth=3.2, origin=(0.0, 0.0, 0.0), delta=0.3, epsilon=0.025)
    K = scaled_K(64, 48)
    oo, ot = make_oracle(m, K, vol)
    oo.create_circle(1.0, 1.6, 1.6, 1.6)
    go, gt = make_gpu(m, K, vol)
    go.upload(oo.D, oo.W)
    eye, zero = np.eye(3), np.zeros(3)
    ot.set_camera_transformation(eye, zero)
    gt.set_camera_transformation(eye, zero)
    ncol, nrow = 40, 30                       # 1200 samples: 5 workgroups of 256
    xyz = np.full((3 * nrow - 2, 3 * ncol - 2, 3), np.nan, dtype=np.float32)
    samp = np.full((ncol, nrow, 3), [-5.0, 1.0, 1.0], dtype=np.float32)      # all out of grid
    samp[0, 5] = (1.7, 1.5, 1.4)               # sample index 5: valid
    samp[0, 7] = (np.nan, 0, 0)                # NaN samples are transparent
    samp[30, 2] = (1.2, 1.9, 1.6)              # index 902: valid, ends the run
    samp[30, 3] = (0.02, 0.02, 0.02)           # in grid, but rotational look-ups fine; keep simple
    for c in range(ncol):
        for r in range(nrow):
            xyz[3 * r, 3 * c] = samp[c, r]
    A_o, b_o, st_o = ot.accumulate(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
    go.set_frame(xyz)
    A_g, b_g, st_g = gt.accumulate()
    assert st_o["n_terms"] > 900
    assert st_g["n_terms"] == st_o["n_terms"] and st_g["n_ok"] == st_o["n_ok"]
    assert sym_rel_err(A_g, A_o) < 1e-11 and sym_rel_err(b_g, b_o) < 1e-11


def test_track_one_frame_pose():
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m, noise=True)
    k = 3
    ot.set_camera_transformation(seq.R[k - 1], seq.t[k - 1])
    gt.set_camera_transformation(seq.R[k - 1], seq.t[k - 1])
    xyz = fr[k][0]
    st_o = ot.estimate_new_position(oo, orc.Cloud(xyz), threads=1, stale_carry=True)
    st_g = gt.estimate_new_position(go, xyz)
    assert not st_o["nonfinite"]
    assert st_g["iterations"] == st_o["iterations"] and bool(st_g["stopped"]) == st_o["stopped"]
    assert np.max(np.abs(gt.rot - ot.rot)) < 1e-9 and np.max(np.abs(gt.trans - ot.trans)) < 1e-9
    assert np.max(np.abs(gt.rot_inv_trans - ot.rot_inv_trans)) < 1e-9
    assert np.allclose(st_g["last_twist"], st_o["last_twist"], atol=1e-9)


def test_free_running_ten_frames():
    """The reference's frame rule (sdf_reconstruction.cpp:69-74): frame 1 integrate only, then
    track -> integrate.  Trajectories must agree to 1e-5 m over 10 frames."""
    m = 64
    seq, fr = frames(10, noise=True, holes=0.01, step=2)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K)
    worst = 0.0
    for k, (xyz, nrm, rgb) in enumerate(fr):
        cloud = orc.Cloud(xyz, nrm, rgb)
        if k > 0:
            so = ot.estimate_new_position(oo, cloud, threads=1, stale_carry=True)
            sg = gt.estimate_new_position(go, xyz)
            assert not so["nonfinite"]
            worst = max(worst, float(np.max(np.abs(gt.trans - ot.trans))), float(np.max(np.abs(gt.rot - ot.rot))))
        oo.update(ot, cloud)
        go.update(gt, xyz, nrm, rgb)
    assert worst < 1e-5, worst
    D, W = go.download()
    # volumes agree except where a last-bit pose difference flipped a pixel truncation
    assert float((ulp_diff(W, oo.W) > 1).mean()) < 1e-3


def test_slab_shards_sum_to_whole():
    """x-slab sharding emulated on one GPU: three handles, each a slab + halo."""
    import tracking_sdf_amd as ts
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m, holes=0.02)
    k = 3
    xyz, nrm, rgb = fr[k]
    pose = (seq.R[k], seq.t[k])
    gt.set_camera_transformation(*pose)
    go.set_frame(xyz)
    A, b, st = gt.accumulate()
    halo = ts.halo_for(go.cfg, 6.0)
    parts = []
    nr = 3
    for r in range(nr):
        x0, x1 = ts.slab_range(m, nr, r)
        gs, gtr = make_gpu(m, seq.K, slab=(x0, x1), halo=halo)
        gs.upload_with_halo(oo.D, oo.W)
        gtr.set_camera_transformation(*pose)
        gs.set_frame(xyz)
        parts.append(gtr.accumulate())
        # a slab integrates its own layers (+ halo) to the same bits as the whole volume
        gs.set_frame(xyz, nrm, rgb)
        sti = gs.update()
        Ds, Ws = gs.download()
        go2, gt2 = make_gpu(m, seq.K)
        go2.upload(oo.D, oo.W)
        gt2.set_camera_transformation(*pose)
        go2.update(gt2, xyz, nrm, rgb)
        Df, Wf = go2.download()
        sl = slice(x0 * m * m, x1 * m * m)
        assert np.array_equal(Ds, Df[sl], equal_nan=True) and np.array_equal(Ws, Wf[sl], equal_nan=True)
        assert sti["n_voxels"] == (min(m, x1 + halo) - max(0, x0 - halo)) * m * m
    assert sum(p[2]["n_terms"] for p in parts) == st["n_terms"]
    assert sum(p[2]["n_ok"] for p in parts) == st["n_ok"]
    assert sym_rel_err(sum(p[0] for p in parts), A) < 1e-12
    assert sym_rel_err(sum(p[1] for p in parts), b) < 1e-12


@pytest.mark.parametrize("m,nr,block,with_color", [(64, 2, 16, True), (64, 4, 8, False), (128, 4, 16, True)])
def test_block_cyclic_shards_sum_to_whole(m, nr, block, with_color):
    """Block-cyclic placement (tsdf_config::slab_stride): rank r owns the blocks [r B + j N B, (r+1) B + j N B),
    each stored with its halo.  Emulated on one GPU, one handle per rank: every rank integrates the same three noisy frames
    at the same poses; its own layers equal the same layers of the whole volume bit for bit (D, W, colour), the owned
    n_updated counts add up to the whole volume's per frame, and one Gauss-Newton pass at a fourth pose adds up -- counts
    exactly, A and b to 1e-12 (f64, another summation order).  tsdf_sample agrees wherever a rank stores the point."""
    import tracking_sdf_amd as ts
    seq, fr = frames(4, noise=True, holes=0.02)
    whole, wt = make_gpu(m, seq.K, with_color=with_color)
    halo = ts.halo_for(whole.cfg, 6.0)
    assert nr * block - block >= 2 * halo or pytest.skip("halo too wide for this block size")
    ranks = [make_gpu(m, seq.K, with_color=with_color, slab=(r * block, (r + 1) * block), halo=halo, slab_stride=nr * block) for r in range(nr)]
    for k in range(3):
        xyz, nrm, rgb = fr[k]
        wt.set_camera_transformation(seq.R[k], seq.t[k])
        st = whole.update(wt, xyz, nrm, rgb if with_color else None)
        own = 0
        for gs, gtr in ranks:
            gtr.set_camera_transformation(seq.R[k], seq.t[k])
            sr = gs.update(gtr, xyz, nrm, rgb if with_color else None)
            own += sr["n_updated"]
        assert own == st["n_updated"] and own > 0
    Df, Wf = (a.reshape(m, m * m) for a in whole.download())
    colf = [a.reshape(m, m * m) for a in whole.download_color()] if with_color else None
    seen = np.zeros(m, bool)
    for gs, _ in ranks:
        xs = gs.owned_x()
        assert not seen[xs].any()
        seen[xs] = True
        Ds, Ws = (a.reshape(len(xs), m * m) for a in gs.download())
        assert np.array_equal(Ds, Df[xs], equal_nan=True) and np.array_equal(Ws, Wf[xs], equal_nan=True)
        if with_color:
            for a, b in zip(gs.download_color(), colf):
                assert np.array_equal(a.reshape(len(xs), m * m), b[xs], equal_nan=True)
    assert seen.all()                                       # every layer has exactly one owner
    # one Gauss-Newton pass at the pose of the fourth frame
    pose = (seq.R[3], seq.t[3])
    xyz = fr[3][0]
    wt.set_camera_transformation(*pose)
    whole.set_frame(xyz)
    A, b, st = wt.accumulate()
    parts = []
    for gs, gtr in ranks:
        gtr.set_camera_transformation(*pose)
        gs.set_frame(xyz)
        parts.append(gtr.accumulate())
    for key in ("n_terms", "n_ok", "n_in_grid_owned"):
        assert sum(p[2][key] for p in parts) == st[key], key
    assert st["n_ok"] > 0.3 * st["n_samples"]
    assert sym_rel_err(sum(p[0] for p in parts), A) < 1e-12
    assert sym_rel_err(sum(p[1] for p in parts), b) < 1e-12
    # device-side sampling: a point is answered by the rank that stores its cell, with the whole volume's value
    rng = np.random.default_rng(3)
    pts = rng.uniform(1.0, m - 2.0, size=(4000, 3))
    vw, okw = whole.interpolate_distance(pts, raw=True)
    answered = np.zeros(len(pts), bool)
    for gs, _ in ranks:
        mine = np.isin(np.floor(pts[:, 0]).astype(int), gs.owned_x())        # the cell starts in one of the rank's own layers
        v, ok = gs.interpolate_distance(pts[mine], raw=True)
        assert np.array_equal(ok, okw[mine]) and np.array_equal(v.view(np.uint32), vw[mine].view(np.uint32))
        answered |= mine
    assert answered.all()
    with pytest.raises(ts.TsdfError) as ei:                # ... and a point two blocks away is refused, not answered from elsewhere
        far = np.array([[float(ranks[0][0].owned_x()[0] + block + halo + 1) + 0.5, m / 2, m / 2]])
        ranks[0][0].interpolate_distance(far)
    assert ei.value.code == ts.E_HALO
    # marching cubes: every rank meshes the cubes whose base layer it owns, block by block; together they are the whole mesh
    def tri_sorted(v, c=None):
        key = v.reshape(len(v), -1).view(np.uint32)
        order = np.lexsort(key.T[::-1])
        return (v[order], c[order]) if c is not None else v[order]
    if with_color:
        vw_, cw_ = whole.mesh(with_color=True)
        got = [gs.mesh(with_color=True) for gs, _ in ranks]
        vr, cr = np.concatenate([g_[0] for g_ in got]), np.concatenate([g_[1] for g_ in got])
        assert len(vw_) > 500 and len(vr) == len(vw_)
        a_, b_ = tri_sorted(vw_, cw_), tri_sorted(vr, cr)
        assert np.array_equal(a_[0].view(np.uint32), b_[0].view(np.uint32)) and np.array_equal(a_[1].view(np.uint32), b_[1].view(np.uint32))
    else:
        vw_ = whole.mesh()
        vr = np.concatenate([gs.mesh() for gs, _ in ranks])
        assert len(vw_) > 500 and np.array_equal(tri_sorted(vw_).view(np.uint32), tri_sorted(vr).view(np.uint32))
    # checkpoints: a rank's own file restores it (every stored layer of every block); the WHOLE volume's file restores any
    # placement; a file of another placement, and a rank's file in a plain handle, are refused
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        whole.save(td + "/whole.vol")
        g0 = ranks[0][0]
        g0.save(td + "/r0.vol")
        want = g0.download() + (tuple(g0.download_color()) if with_color else ())
        for src in ("/r0.vol", "/whole.vol"):
            fresh, _ = make_gpu(m, seq.K, with_color=with_color, slab=(0, block), halo=halo, slab_stride=nr * block)
            fresh.load(td + src)
            have = fresh.download() + (tuple(fresh.download_color()) if with_color else ())
            assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(want, have))
            # ... halos included: the restored handle tracks like the one that was saved
            trk = ts.CameraTracking(20, 0.001, 1.0, 0.01, fresh)
            trk.set_K(seq.K)
            trk.set_camera_transformation(*pose)
            fresh.set_frame(xyz)
            A1, b1, st1 = trk.accumulate()
            assert np.array_equal(A1, parts[0][0]) and np.array_equal(b1, parts[0][1]) and st1 == parts[0][2]
            fresh.close()
        with pytest.raises(ts.TsdfError):
            ranks[1][0].load(td + "/r0.vol")                  # rank 0's blocks are not rank 1's
        with pytest.raises(ts.TsdfError):
            whole.load(td + "/r0.vol")
    for gs, _ in ranks:
        gs.close()
    whole.close()


def test_block_cyclic_configurations_that_are_refused():
    import tracking_sdf_amd as ts
    for kw in (dict(m=96, slab=(0, 16), halo=2, slab_stride=32),        # m not a power of two
               dict(m=64, slab=(0, 12), halo=2, slab_stride=24),        # m not a multiple of the block
               dict(m=64, slab=(8, 24), halo=2, slab_stride=32),        # block not at a multiple of its size
               dict(m=64, slab=(0, 16), halo=9, slab_stride=32),        # stored ranges of two blocks would overlap
               dict(m=64, slab=(0, 16), halo=2, slab_stride=-16)):
        with pytest.raises(ts.TsdfError):
            ts.SDF(with_color=False, **kw)


def test_halo_too_small_is_reported():
    import tracking_sdf_amd as ts
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m)
    gs, gtr = make_gpu(m, seq.K, slab=(20, 40), halo=0)
    gs.upload_with_halo(oo.D, oo.W)
    gtr.set_camera_transformation(seq.R[3], seq.t[3])
    gs.set_frame(fr[3][0])
    with pytest.raises(ts.TsdfError) as ei:
        gtr.accumulate()
    assert ei.value.code == ts.E_HALO


def test_error_paths_leave_pose_untouched():
    import tracking_sdf_amd as ts
    m = 32
    seq, fr = frames(1)
    go = ts.SDF(m)
    gt = ts.CameraTracking(sdf=go)
    xyz, nrm, rgb = fr[0]
    with pytest.raises(ts.TsdfError) as ei:
        go.update()                                   # no frame
    assert ei.value.code == ts.E_NO_FRAME
    go.set_frame(xyz, nrm, rgb)
    with pytest.raises(ts.TsdfError) as ei:
        go.update()                                   # K never set: the reference exits here
    assert ei.value.code == ts.E_NO_INTRINSICS
    rot0, trans0 = gt.rot.copy(), gt.trans.copy()
    with pytest.raises(ts.TsdfError) as ei:
        gt.estimate_new_position(go, xyz)             # empty volume: no valid sample
    assert ei.value.code == ts.E_NO_SAMPLES
    assert np.array_equal(gt.rot, rot0) and np.array_equal(gt.trans, trans0)
    with pytest.raises(ts.TsdfError) as ei:
        gt.gn_update(np.zeros((6, 6)), np.ones(6))    # singular: the reference would go NaN silently
    assert ei.value.code == ts.E_SINGULAR
    assert np.array_equal(gt.rot, rot0) and np.array_equal(gt.trans, trans0)
    go.set_frame(xyz)                                 # xyz only: integrate must refuse
    gt.set_K(seq.K)
    with pytest.raises(ts.TsdfError):
        go.update()


@pytest.mark.parametrize("fault,code", [("no_samples", "E_NO_SAMPLES"), ("singular", "E_SINGULAR"), ("comm", "E_COMM"),
                                        ("halo", "E_HALO")])
def test_failure_in_a_later_pass_restores_the_entry_pose(fault, code):
    """tsdf.h: 'pose left unchanged' must also hold when pass g > 0 fails, after passes 0..g-1 moved the pose.
    The per-pass all-reduce hook injects the failure at the second pass (first pass solved and applied)."""
    import tracking_sdf_amd as ts
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m, noise=True)
    xyz = fr[3][0]
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    rot0, trans0 = gt.rot.copy(), gt.trans.copy()
    st = gt.estimate_new_position(go, xyz)
    assert st["iterations"] >= 2                           # otherwise the fault below would never be reached
    assert not np.array_equal(gt.trans, trans0)
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    calls = []

    def hook(arr):
        calls.append(1)
        if len(calls) < 2:
            return
        if fault == "no_samples":
            arr[:] = 0.0                                   # n_terms (entry 27) = 0
        elif fault == "singular":
            arr[:21] = 0.0                                 # A = 0, terms > 0
        elif fault == "halo":
            arr[28] = 3.0                                  # look-ups outside the stored layers, summed over ranks
        else:
            raise RuntimeError("collective failed")
    go.set_allreduce_hook(hook)
    with pytest.raises(ts.TsdfError) as ei:
        gt.estimate_new_position(go, xyz)
    assert ei.value.code == getattr(ts, code)
    assert len(calls) == 2
    assert np.array_equal(gt.rot, rot0) and np.array_equal(gt.trans, trans0)
    go.set_allreduce_hook(None)
    st2 = gt.estimate_new_position(go, xyz)                # and the handle is still usable: same result as before
    assert st2["iterations"] == st["iterations"]


def test_download_upload_roundtrip_and_reset():
    import tracking_sdf_amd as ts
    m = 32
    go = ts.SDF(m)
    D, W = go.download()
    assert np.all(D == np.float32(15.5)) and np.all(W == 0)
    cw, r, g, b = go.download_color()
    assert np.all(cw == 0) and np.all(r == np.float32(0.4)) and np.all(b == np.float32(0.4))
    rng = np.random.default_rng(0)
    D2 = rng.standard_normal(m ** 3).astype(np.float32)
    W2 = rng.random(m ** 3).astype(np.float32)
    go.upload(D2, W2)
    D3, W3 = go.download()
    assert np.array_equal(D3, D2) and np.array_equal(W3, W2)
    go.reset()
    D, W = go.download()
    assert np.all(D == np.float32(15.5)) and np.all(W == 0)


def test_rccl_single_rank_allreduce_and_hook():
    """The in-library RCCL path (dlopen'ed librccl) with a 1-rank communicator, and the host hook."""
    import ctypes
    import tracking_sdf_amd as ts
    go = ts.SDF(32)
    buf = ctypes.create_string_buffer(128)
    assert ts.lib().tsdf_comm_unique_id(buf) == 0
    go.comm_init(1, 0, buf.raw)
    v = np.arange(30, dtype=np.float64) * 0.25
    assert np.array_equal(go.allreduce(v.copy()), v)
    go.comm_finalize()
    calls = []

    def hook(arr):
        calls.append(arr.size)
        arr *= 2.0            # pretend a second rank holds the same partial sums
    go.set_allreduce_hook(hook)
    assert np.array_equal(go.allreduce(v.copy()), 2 * v)
    assert calls == [30]
    go.set_allreduce_hook(None)
    assert np.array_equal(go.allreduce(v.copy()), v)


def test_track_with_hook_doubles_normal_equations_consistently():
    """A hook that doubles A and b (two identical ranks) must leave the solution, hence the pose, unchanged."""
    m = 64
    seq, fr, oo, ot, go, gt = _fused_pair(m, noise=True)
    xyz = fr[3][0]
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    st1 = gt.estimate_new_position(go, xyz)
    rot1, trans1 = gt.rot.copy(), gt.trans.copy()

    def hook(arr):
        arr *= 2.0
    go.set_allreduce_hook(hook)
    gt.set_camera_transformation(seq.R[2], seq.t[2])
    st2 = gt.estimate_new_position(go, xyz)
    assert st2["iterations"] == st1["iterations"] and st2["n_terms_last"] == 2 * st1["n_terms_last"]
    assert np.max(np.abs(gt.rot - rot1)) < 1e-12 and np.max(np.abs(gt.trans - trans1)) < 1e-12


@pytest.mark.parametrize("roll_deg", [90.0, 35.0])
def test_integrate_rolled_camera_uses_other_record_layout(roll_deg):
    """A camera rolled about its optical axis maps voxel k-rows to image ROWS: the packed pixel records
    switch to row-major order.  Results must not depend on the layout."""
    m = 64
    seq, fr = frames(1)
    a = np.deg2rad(roll_deg)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    R = seq.R[0] @ Rz
    xyz, nrm, rgb = synth.render_frame(R, seq.t[0], seq.K, W_, H_, noise=True, holes=0.02,
                                       rng=np.random.default_rng(5))
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K)
    ot.set_camera_transformation(R, seq.t[0])
    gt.set_camera_transformation(R, seq.t[0])
    n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    st = go.update(gt, xyz, nrm, rgb)
    assert st["n_updated"] == n_or and n_or > 1000
    assert_volume_equal(go, oo, m)


def test_integrate_general_intrinsics_disable_row_clip():
    """A K whose last row is not (0,0,1) turns the per-row frustum clip off; results still match."""
    m = 32
    seq, fr = frames(1)
    K = seq.K.copy()
    K[2] = [1e-4, -2e-4, 1.0]
    K[0, 1] = 0.3
    oo, ot = make_oracle(m, K)
    go, gt = make_gpu(m, K)
    xyz, nrm, rgb = fr[0]
    n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
    st = go.update(gt, xyz, nrm, rgb)
    assert st["n_updated"] == n_or
    assert_volume_equal(go, oo, m)


def test_checkpoint_roundtrip(tmp_path):
    """tsdf_save / tsdf_load: the volume survives a process-independent file, bit for bit."""
    import tracking_sdf_amd as ts
    m = 32
    seq, fr = frames(2)
    go, gt = make_gpu(m, seq.K)
    go.update(gt, *fr[0])
    path = str(tmp_path / "vol.tsdf")
    go.save(path)
    D, W = go.download()
    col = go.download_color()
    assert os.path.getsize(path) == 80 + 6 * 4 * m ** 3
    go2, gt2 = make_gpu(m, seq.K)
    go2.load(path)
    D2, W2 = go2.download()
    assert np.array_equal(D, D2) and np.array_equal(W, W2)
    for a, b in zip(col, go2.download_color()):
        assert np.array_equal(a, b)
    # resuming from the checkpoint continues exactly like the original
    gt.set_camera_transformation(seq.R[1], seq.t[1]); gt2.set_camera_transformation(seq.R[1], seq.t[1])
    go.update(gt, *fr[1]); go2.update(gt2, *fr[1])
    assert all(np.array_equal(a, b) for a, b in zip(go.download(), go2.download()))
    # mismatching handle is refused
    go3 = ts.SDF(m, with_color=False)
    with pytest.raises(ts.TsdfError):
        go3.load(path)
    with pytest.raises(ts.TsdfError):
        go2.load(str(tmp_path / "missing.tsdf"))


def test_sharded_checkpoint_restores_halo_and_tracks_like_the_whole(tmp_path):
    """A shard restored from a checkpoint must hold its halo layers too (they are the neighbour's interior), so that
    its tracker partial sums add up to the whole volume's -- from the shard's own file and from a whole-volume file;
    a file that does not cover the stored layers, or was written for another geometry, is refused."""
    import tracking_sdf_amd as ts
    m, halo = 48, 6
    seq, fr = frames(4)
    whole, wt = make_gpu(m, seq.K)
    shards = [make_gpu(m, seq.K, slab=ts.slab_range(m, 2, r), halo=halo) for r in range(2)]
    for k in range(3):                                     # the same frames into the whole volume and both shards
        for s_, t_ in [(whole, wt)] + shards:
            t_.set_camera_transformation(seq.R[k], seq.t[k])
            s_.update(t_, *fr[k])
    wpath = str(tmp_path / "whole.tsdf")
    whole.save(wpath)
    wt.set_camera_transformation(seq.R[2], seq.t[2])
    whole.set_frame(fr[3][0])
    A, b, st = wt.accumulate()
    for source in ("own", "whole"):
        As, bs, terms = 0.0, 0.0, 0
        for r, (s_, t_) in enumerate(shards):
            path = wpath if source == "whole" else str(tmp_path / f"shard{r}.tsdf")
            if source == "own":
                s_.save(path)
                x0, x1 = ts.slab_range(m, 2, r)
                xs, xe = max(0, x0 - halo), min(m, x1 + halo)
                assert os.path.getsize(path) == 80 + 6 * 4 * (xe - xs) * m * m      # slab AND halo
            fresh, ft = make_gpu(m, seq.K, slab=ts.slab_range(m, 2, r), halo=halo)
            fresh.load(path)
            assert all(np.array_equal(a_, b_) for a_, b_ in zip(fresh.download(), s_.download()))
            assert all(np.array_equal(a_, b_) for a_, b_ in zip(fresh.download_color(), s_.download_color()))
            ft.set_camera_transformation(seq.R[2], seq.t[2])
            fresh.set_frame(fr[3][0])
            Ar, br, sr = ft.accumulate()                  # would raise E_HALO or give other sums with a stale halo
            As, bs, terms = As + Ar, bs + br, terms + sr["n_terms"]
            # the halo layers came back as well: one more frame keeps the shard identical to the never-saved one
            ft.set_camera_transformation(seq.R[3], seq.t[3]); t_.set_camera_transformation(seq.R[3], seq.t[3])
            fresh.update(ft, *fr[3])
            probe, pt = make_gpu(m, seq.K, slab=ts.slab_range(m, 2, r), halo=halo)
            probe.load(path)
            pt.set_camera_transformation(seq.R[3], seq.t[3])
            probe.update(pt, *fr[3])
            assert all(np.array_equal(a_, b_) for a_, b_ in zip(fresh.download(), probe.download()))
        assert terms == st["n_terms"]
        assert sym_rel_err(As, A) < 1e-12 and sym_rel_err(bs, b) < 1e-12
    # shard 0's file does not cover shard 1's layers
    other, _ = make_gpu(m, seq.K, slab=ts.slab_range(m, 2, 1), halo=halo)
    with pytest.raises(ts.TsdfError) as ei:
        other.load(str(tmp_path / "shard0.tsdf"))
    assert ei.value.code == ts.E_HALO
    # same m, other truncation distance: the stored distances would mean something else
    vol2 = dict(VOL, delta=0.2)
    alien, _ = make_gpu(m, seq.K, vol=vol2)
    with pytest.raises(ts.TsdfError) as ei:
        alien.load(wpath)
    assert ei.value.code == ts.E_BADARG


def test_track_and_integrate_equals_the_two_calls():
    """tsdf_track_and_integrate = tsdf_track followed by tsdf_integrate (same pose, same volume), and it integrates
    nothing when tracking fails."""
    import ctypes as C
    import tracking_sdf_amd as ts
    m = 48
    seq = synth.Sequence(n_frames=3, width=160, height=120, noise=True, holes=0.02, step=3)
    frames = [seq.frame(k) for k in range(3)]
    out = []
    for fused in (False, True):
        s, t = make_gpu(m, seq.K)
        s.update(t, *frames[0])
        for k in (1, 2):
            s.set_frame(*frames[k])
            if fused:
                s._check(ts.lib().tsdf_track_and_integrate(s._h, 1, None, None))
            else:
                t.estimate_new_position()
                s.update()
        out.append((t.rot.copy(), t.trans.copy()) + s.download())
        if fused:
            nan = np.full_like(frames[0][0], np.nan)
            s.set_frame(nan, frames[0][1], frames[0][2])
            assert ts.lib().tsdf_track_and_integrate(s._h, 1, None, None) == ts.E_NO_SAMPLES
            D, W = s.download()
            assert np.array_equal(W, out[-1][3])              # nothing was integrated after the failed track
        s.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_work_list_regions_follow_the_previous_frame_and_overflow_when_it_lies():
    """The integrate work list is placed in per-band regions sized from the PREVIOUS launch's band counts, with an
    overflow region for whatever does not fit (list_rows_kernel).  Drive the prediction wrong on purpose: a camera that
    alternates between looking along the path, rolled by 90 degrees (other record layout, other bands) and standing
    50 m away (the whole volume projects into a few pixels of ONE band, nothing is updated), 8 launches on one handle.  Every
    launch must update exactly the oracle's voxels and the volume must come out bit-identical."""
    m = 96
    seq, fr = frames(8, noise=True, holes=0.02, step=9)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K)
    a = np.deg2rad(90.0)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    counts = []
    for k in range(8):
        R = seq.R[k] @ (Rz if k % 4 == 1 else np.eye(3))
        t = seq.t[k] - (50.0 * R[:, 2] if k % 4 == 2 else 0.0)                # backwards along the optical axis
        xyz, nrm, rgb = synth.render_frame(R, t, seq.K, W_, H_, noise=True, holes=0.02, rng=np.random.default_rng(40 + k))
        ot.set_camera_transformation(R, t)
        gt.set_camera_transformation(R, t)
        n_or = oo.update(ot, orc.Cloud(xyz, nrm, rgb))
        st = go.update(gt, xyz, nrm, rgb)
        assert st["n_updated"] == n_or, (k, st["n_updated"], n_or)
        counts.append(n_or)
    assert max(counts) > 20 * (min(counts) + 1), counts          # the launches really differ
    assert_volume_equal(go, oo, m)


def test_unnormalised_normals_in_the_exp_band():
    """sdf.cpp:294 divides by n.norm(): the input normals need not be unit vectors.  Their length enters the distance
    (d = (P - voxel) . n), so a normal far from 1 moves its pixel's voxels out of the exp() band -- lengths within a
    factor of a few keep some in it.  The band's colour weight (float)(w * |n_z| / |n|) is computed by the bare
    square-root / division cores for normals of ordinary size and by sqrt() and '/' otherwise: both against the
    oracle's libm, three frames, lengths from 2^-4 to 2^4 with a sprinkling of 2^+-30, 2^+-120, zero vectors and
    infinite components."""
    m = 48
    seq, fr = frames(3, noise=True, holes=0.01)
    rng = np.random.default_rng(11)
    oo, ot = make_oracle(m, seq.K)
    go, gt = make_gpu(m, seq.K)
    for xyz, nrm, rgb in fr:
        scale = np.exp2(rng.uniform(-4.0, 4.0, size=nrm.shape[:2])).astype(np.float32)
        odd = rng.random(nrm.shape[:2])
        scale[odd < 0.02] = np.float32(2.0 ** 30)
        scale[(odd >= 0.02) & (odd < 0.04)] = np.float32(2.0 ** -30)
        scale[(odd >= 0.04) & (odd < 0.05)] = np.float32(2.0 ** 120)
        scale[(odd >= 0.05) & (odd < 0.06)] = np.float32(2.0 ** -120)
        n2 = (nrm * scale[..., None]).astype(np.float32)
        n2[(odd >= 0.06) & (odd < 0.07)] = 0.0
        n2[(odd >= 0.07) & (odd < 0.08), 2] = np.inf
        n2[(odd >= 0.08) & (odd < 0.09), 0] = -np.inf
        with np.errstate(all="ignore"):
            n_or = oo.update(ot, orc.Cloud(xyz, n2, rgb), with_color=True)
        st = go.update(gt, xyz, n2, rgb)
        assert st["n_updated"] == n_or
    D, W = go.download()
    uW = ulp_diff(W, oo.W)
    assert uW.max() <= 1 and ulp_diff(D, oo.D)[uW == 0].max() == 0
    cw, r, g, b = go.download_color()
    same = uW == 0
    for got, want in ((cw, oo.Color_W), (r, oo.R), (g, oo.G), (b, oo.B)):
        a, w = got[same], want[same]
        assert np.array_equal(np.isnan(a), np.isnan(w))
        ok = ~np.isnan(w)
        assert np.array_equal(a[ok].view(np.uint32), w[ok].view(np.uint32))


@pytest.mark.parametrize("per_cu,m", [(1, 40), (3, 64), (7, 64), (2, 96)])
def test_the_volume_does_not_depend_on_the_integrate_grid(per_cu, m, monkeypatch):
    """The workgroups of an XCD walk its part of the work list together (workgroup w takes the items
    [(j * per_xcd + w) * 4, +4)), the XCD shares follow the last launch's timers: every voxel belongs to exactly one
    item, so the volume must not depend on how many workgroups there are -- 1, 2, 3 or 7 per CU instead of 5, over
    four frames (the feedback has moved the shares by then), against the default grid bit for bit."""
    seq, fr = frames(4, noise=True, holes=0.02)

    def run():
        go, gt = make_gpu(m, seq.K)
        n = [go.update(gt, *f)["n_updated"] for f in fr]
        out = (n, go.download(), go.download_color())
        go.close()
        return out
    monkeypatch.delenv("TSDF_INTEGRATE_BLOCKS_PER_CU", raising=False)
    want = run()
    monkeypatch.setenv("TSDF_INTEGRATE_BLOCKS_PER_CU", str(per_cu))
    got = run()
    assert want[0] == got[0]
    for a, b in zip(want[1] + want[2], got[1] + got[2]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
