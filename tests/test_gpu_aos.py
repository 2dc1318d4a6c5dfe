"""tsdf_set_frame_aos: the frame in the reference's own format (arrays of PCL point structs) against the planar entry
point -- bit-identical volumes and poses, including the two-call sequence of the reference (points for the tracker,
then the normals of the same cloud for the integration)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tracking_sdf_amd import synth

pytestmark = pytest.mark.gpu

W, H, M = 160, 120, 48


def clouds(xyz, nrm, rgb, point_dtype=None, normal_dtype=None):
    import tracking_sdf_amd as ts
    pts = np.zeros(xyz.shape[:2], dtype=point_dtype or ts.PCL_POINT_XYZRGB)
    pts["x"], pts["y"], pts["z"] = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    if "r" in pts.dtype.fields:
        pts["r"], pts["g"], pts["b"] = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    nn = np.zeros(xyz.shape[:2], dtype=normal_dtype or ts.PCL_NORMAL)
    nn["normal_x"], nn["normal_y"], nn["normal_z"] = nrm[..., 0], nrm[..., 1], nrm[..., 2]
    return pts, nn


def run_sequence(feed, n=4, color=True):
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    s = ts.SDF(M, with_color=color)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    poses = []
    for k in range(n):
        feed(s, t, k, *seq.frame(k))
        poses.append((t.rot.copy(), t.trans.copy()))
    D, Wt = s.download()
    col = s.download_color() if color else None
    s.close()
    return poses, D, Wt, col


def planar(s, t, k, xyz, nrm, rgb):
    if k > 0:
        s.set_frame(xyz, None, rgb)
        t.estimate_new_position()
    s.set_frame(xyz, nrm, rgb)
    s.update()


def aos_reference_order(s, t, k, xyz, nrm, rgb):
    pts, nn = clouds(xyz, nrm, rgb)
    if k > 0:
        s.set_frame_aos(pts)                  # estimate_new_position(sdf, cloud_filtered)
        t.estimate_new_position()
        s.set_frame_aos(None, nn)             # update(tracker, cloud_filtered, normals): the points are in HBM already
    else:
        s.set_frame_aos(pts, nn)
    s.update()


def aos_tight(s, t, k, xyz, nrm, rgb):
    """another layout: 16-byte points with the colour bytes in r, g, b order, 12-byte normals"""
    pd = np.dtype({"names": ["r", "g", "b", "x", "y", "z"], "formats": ["u1", "u1", "u1", "<f4", "<f4", "<f4"],
                   "offsets": [0, 1, 2, 4, 8, 12], "itemsize": 16})
    nd = np.dtype([("normal_x", "<f4"), ("normal_y", "<f4"), ("normal_z", "<f4")])
    pts, nn = clouds(xyz, nrm, rgb, pd, nd)
    if k > 0:
        s.set_frame_aos(pts)
        t.estimate_new_position()
    s.set_frame_aos(pts, nn)
    s.update()


def aos_entry_points(s, t, k, xyz, nrm, rgb):
    """tsdf_track_aos / tsdf_integrate_aos: the reference's two calls, the cloud handed to both (sdf_reconstruction.cpp:70,74)"""
    pts, nn = clouds(xyz, nrm, rgb)
    if k > 0:
        s.track_aos(pts)                      # estimate_new_position(sdf, cloud_filtered): samples first
    s.update_aos(pts, nn)                     # update(tracker, cloud_filtered, normals); k == 0: nothing was tracked


def aos_entry_points_no_cloud(s, t, k, xyz, nrm, rgb):
    pts, nn = clouds(xyz, nrm, rgb)
    if k > 0:
        s.track_aos(pts)
        s.update_aos(None, nn)                # "the tracked cloud"
    else:
        s.update_aos(pts, nn)


def aos_entry_points_whole_frame(s, t, k, xyz, nrm, rgb):
    """tsdf_track_frame_aos: the normals handed over at tracking time, the whole frame staged and packed under the passes;
    plain tsdf_integrate then integrates what was staged"""
    pts, nn = clouds(xyz, nrm, rgb)
    if k > 0:
        s.track_aos(pts, nn)
        s.update()
    else:
        s.update_aos(pts, nn)


def aos_entry_points_whole_frame_checked(s, t, k, xyz, nrm, rgb):
    """... and tsdf_integrate_aos behind it: both clouds compared with what was staged, nothing uploaded again"""
    pts, nn = clouds(xyz, nrm, rgb)
    if k > 0:
        s.track_aos(pts, nn)
    s.update_aos(pts if k % 2 else None, nn) if k > 0 else s.update_aos(pts, nn)


def aos_entry_points_tight(s, t, k, xyz, nrm, rgb):
    pd = np.dtype({"names": ["r", "g", "b", "x", "y", "z"], "formats": ["u1", "u1", "u1", "<f4", "<f4", "<f4"],
                   "offsets": [0, 1, 2, 4, 8, 12], "itemsize": 16})
    nd = np.dtype([("normal_x", "<f4"), ("normal_y", "<f4"), ("normal_z", "<f4")])
    pts, nn = clouds(xyz, nrm, rgb, pd, nd)
    if k > 0:
        s.track_aos(pts)
    s.update_aos(pts, nn)


@pytest.mark.parametrize("feed", [aos_reference_order, aos_tight, aos_entry_points, aos_entry_points_no_cloud, aos_entry_points_tight,
                                  aos_entry_points_whole_frame, aos_entry_points_whole_frame_checked])
def test_aos_frames_give_the_planar_result(feed):
    want = run_sequence(planar)
    got = run_sequence(feed)
    for (r0, t0), (r1, t1) in zip(want[0], got[0]):
        assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
    assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])
    for a, b in zip(want[3], got[3]):
        assert np.array_equal(a, b)


def test_points_without_colour_and_argument_checks():
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=1, width=W, height=H, step=4)
    xyz, nrm, rgb = seq.frame(0)
    pd = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4")])
    pts, nn = clouds(xyz, nrm, rgb, pd)
    a = ts.SDF(M, with_color=False); ta = ts.CameraTracking(sdf=a); ta.set_K(seq.K)
    b = ts.SDF(M, with_color=False); tb = ts.CameraTracking(sdf=b); tb.set_K(seq.K)
    a.set_frame_aos(pts, nn); a.update()
    b.set_frame(xyz, nrm); b.update()
    assert all(np.array_equal(x, y) for x, y in zip(a.download(), b.download()))
    c = ts.SDF(M)                                   # with colour: a colourless cloud cannot be integrated
    tc = ts.CameraTracking(sdf=c); tc.set_K(seq.K)
    with pytest.raises(ts.TsdfError) as ei:
        c.set_frame_aos(None, nn)                   # no current host frame to complete
    assert ei.value.code == ts.E_NO_FRAME
    c.set_frame_aos(pts, nn)
    with pytest.raises(ts.TsdfError) as ei:
        c.update()
    assert ei.value.code == ts.E_NO_FRAME
    lay = ts.AosLayout(8, 0, -1, -1, -1, 12, 0)     # 8-byte stride cannot hold three floats
    import ctypes as C
    rc = ts.lib().tsdf_set_frame_aos(c._h, C.c_void_p(pts.ctypes.data), None, C.byref(lay), W, H)
    assert rc == ts.E_BADARG
    small = np.zeros((H // 2, W // 2), dtype=ts.PCL_NORMAL)
    with pytest.raises(ts.TsdfError) as ei:
        c.set_frame_aos(None, small)                # size differs from the current frame
    assert ei.value.code == ts.E_NO_FRAME
    for s in (a, b, c):
        s.close()


def test_staging_threads_chunks_and_sets_do_not_change_the_result():
    """the pageable path's knobs (pool size, chunked copies, one or two staging sets, the tracker's samples uploaded ahead of
    the planes or written by the packing kernel) move bytes, never values: an odd
    image size (planes that are no multiple of 16 bytes, a block whose planes need padding), set one by one and queued"""
    code = ("import numpy as np, tracking_sdf_amd as ts\n"
            "from tracking_sdf_amd import synth\n"
            "seq = synth.Sequence(n_frames=4, width=203, height=117, noise=True, holes=0.02, step=4)\n"
            "s = ts.SDF(32); t = ts.CameraTracking(sdf=s); t.set_K(seq.K)\n"
            "for k in range(2):\n"
            "    s.set_frame(*seq.frame(k))\n"
            "    if k: t.estimate_new_position()\n"
            "    s.update()\n"
            "fr = [tuple(np.ascontiguousarray(a) for a in seq.frame(k)) for k in range(4)]\n"
            "s.queue_frame(*fr[2])\n"
            "for k in (2, 3):\n"
            "    s.next_frame()\n"
            "    if k == 2: s.queue_frame(*fr[3])\n"
            "    t.estimate_new_position(); s.update()\n"
            "D, W = s.download(); print(float(D.sum()), float(W.sum()), int((W > 0).sum()), t.trans.tolist())\n")
    outs = []
    for knobs in ({"TSDF_HOST_THREADS": "1"}, {"TSDF_HOST_THREADS": "3"}, {"TSDF_HOST_THREADS": "7", "TSDF_STAGE_CHUNKS": "4"},
                  {"TSDF_STAGE_CHUNKS": "3"}, {"TSDF_SAMPLES_FIRST": "0"}, {"TSDF_SAMPLES_FIRST": "0", "TSDF_HOST_THREADS": "2"}):
        env = dict(os.environ, **knobs)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                                   cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip())
    assert outs[0] and all(o == outs[0] for o in outs), outs


def run_queued(kind, n=5):
    """The frame loop with the two-deep queue: frame k+1 is queued before frame k is tracked and integrated."""
    import torch
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    frames = []
    for k in range(n):
        xyz, nrm, rgb = seq.frame(k)
        if kind == "pinned":
            hold = [torch.from_numpy(a.copy()).pin_memory() for a in (xyz, nrm, rgb)]
            frames.append(tuple(t.numpy() for t in hold) + (hold,))
        elif kind == "aos":
            frames.append(clouds(xyz, nrm, rgb))
        elif kind == "device":
            hold = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (xyz, nrm, rgb)]
            frames.append(hold)
        else:
            frames.append((np.ascontiguousarray(xyz), np.ascontiguousarray(nrm), np.ascontiguousarray(rgb)))
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)

    def queue(k):
        if kind == "aos":
            s.queue_frame_aos(*frames[k])
        elif kind == "device":
            s.queue_frame_device(frames[k][0].data_ptr(), frames[k][1].data_ptr(), frames[k][2].data_ptr(), W, H, keep=frames[k])
        else:
            s.queue_frame(*frames[k][:3])
    poses = []
    queue(0)
    for k in range(n):
        s.next_frame()                            # frame k becomes current
        if k + 1 < n:
            queue(k + 1)                          # ... and frame k+1 is staged while k is tracked and integrated
            if kind == "device":
                with pytest.raises(ts.TsdfError):     # a frame in device memory waits in the first place only
                    queue(k + 1)
            if k == 1:
                # tsdf_set_frame* while a frame is queued: refused BEFORE any side effect (round 3 re-allocated the frame
                # buffers for the other size and wrote into the planes the staging thread was filling, then refused)
                big = np.zeros((2 * H, 2 * W, 3), np.float32)
                with pytest.raises(ts.TsdfError):
                    s.set_frame(big, big, np.zeros((2 * H, 2 * W, 3), np.uint8))
                with pytest.raises(ts.TsdfError):
                    s.set_frame_aos(*clouds(big, big, np.zeros((2 * H, 2 * W, 3), np.uint8)))
                with pytest.raises(ts.TsdfError):
                    s.set_depth_frame(np.ones((2 * H, 2 * W), np.float32), np.zeros((2 * H, 2 * W, 3), np.uint8))
        if k > 0:
            t.estimate_new_position()
        s.update()
        poses.append((t.rot.copy(), t.trans.copy()))
    with pytest.raises(ts.TsdfError):
        s.next_frame()                            # nothing queued
    D, Wt = s.download()
    col = s.download_color()
    s.close()
    return poses, D, Wt, col


@pytest.mark.parametrize("kind", ["pageable", "pinned", "aos", "device"])
def test_queued_frames_give_the_same_trajectory_and_volume(kind):
    """tsdf_queue_frame / tsdf_next_frame: the next frame is uploaded under the current frame's tracker passes and
    integration; poses and volume must equal the plain set_frame loop bit for bit."""
    want = run_sequence(planar, n=5)
    got = run_queued(kind, n=5)
    for (r0, t0), (r1, t1) in zip(want[0], got[0]):
        assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
    assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])
    for a, b in zip(want[3], got[3]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kinds", [("pageable", "pageable"), ("pinned", "aos"), ("aos", "pageable"), ("device", "pageable")])
def test_two_frames_waiting_give_the_same_trajectory_and_volume(kinds):
    """Two frames behind the current one (frame k+2 is queued while frame k is current): host frames of any kind in either
    place, a frame in device memory in the first place only; a third queued frame is refused, and so is a frame of another
    size.  Poses and volume equal the plain set_frame loop bit for bit."""
    import torch
    import tracking_sdf_amd as ts
    n = 6
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    raw = [seq.frame(k) for k in range(n)]
    held = []

    def queue(s, k, kind):
        xyz, nrm, rgb = (np.ascontiguousarray(a) for a in raw[k])
        if kind == "pinned":
            hold = [torch.from_numpy(a.copy()).pin_memory() for a in (xyz, nrm, rgb)]
            held.append(hold)
            s.queue_frame(*[t.numpy() for t in hold])
        elif kind == "aos":
            s.queue_frame_aos(*clouds(xyz, nrm, rgb))
        elif kind == "device":
            hold = [torch.from_numpy(a).cuda() for a in (xyz, nrm, rgb)]
            held.append(hold)
            s.queue_frame_device(hold[0].data_ptr(), hold[1].data_ptr(), hold[2].data_ptr(), W, H, keep=hold)
        else:
            s.queue_frame(xyz, nrm, rgb)

    want = run_sequence(planar, n=n)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    kind_of = lambda k: kinds[k % 2]
    queue(s, 0, kind_of(0))
    queue(s, 1, kind_of(1))
    poses = []
    for k in range(n):
        s.next_frame()                                    # frame k is current, frame k+1 waits
        if k + 2 < n:
            # a frame in device memory is taken in the first place only: it goes in when nothing else waits, i.e. here it is
            # refused (frame k+1 waits), and the frame goes through host memory instead
            kd = kind_of(k + 2)
            if kd == "device":
                with pytest.raises(ts.TsdfError):
                    queue(s, k + 2, "device")
                kd = "pageable"
            queue(s, k + 2, kd)
            with pytest.raises(ts.TsdfError):             # current + 2: full
                queue(s, k + 2, "pageable")
            if k == 0:
                big = np.zeros((2 * H, 2 * W, 3), np.float32)
                with pytest.raises(ts.TsdfError):
                    s.set_frame(big, big, np.zeros((2 * H, 2 * W, 3), np.uint8))
        if k > 0:
            t.estimate_new_position()
        s.update()
        poses.append((t.rot.copy(), t.trans.copy()))
    with pytest.raises(ts.TsdfError):
        s.next_frame()
    D, Wt = s.download()
    col = s.download_color()
    s.close()
    for (r0, t0), (r1, t1) in zip(want[0], poses):
        assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
    assert np.array_equal(want[1], D) and np.array_equal(want[2], Wt)
    for a_, b_ in zip(want[3], col):
        assert np.array_equal(a_, b_)


@pytest.mark.parametrize("route", ["planes", "clouds"])
def test_synchronize_between_a_host_frame_and_its_update_waits_for_the_planes(route):
    """tsdf_synchronize packs a frame whose packing is still deferred.  For a host frame handed over "samples first" the
    planes may still be on their way when it is called (only tsdf_integrate used to wait for them): at 640x480 the copy takes
    155 us, and a pack launched in front of it read the block's previous content (found by the state-machine walk once
    tsdf_set_frame* had lost 120 us of host time).  Same volume as without the call, bit for bit, ten times over."""
    import tracking_sdf_amd as ts
    w, h, m = 640, 480, 64
    seq = synth.Sequence(n_frames=3, width=w, height=h, noise=True, holes=0.02, step=6)
    frames = [tuple(np.ascontiguousarray(a) for a in seq.frame(k)) for k in range(3)]
    if route == "clouds":
        frames = [clouds(*f) for f in frames]

    def run(sync):
        s = ts.SDF(m, with_color=True)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        for k, f in enumerate(frames):
            t.set_camera_transformation(seq.R[k], seq.t[k])
            s.set_frame(*f) if route == "planes" else s.set_frame_aos(*f)
            if sync:
                s.synchronize()
            s.update(want_stats=False)
        out = s.download() + tuple(s.download_color())
        s.close()
        return out
    want = run(False)
    for _ in range(10):
        got = run(True)
        for a_, b_ in zip(want, got):
            assert np.array_equal(a_, b_)


def test_a_sample_list_sent_ahead_is_not_overwritten_by_an_earlier_frames_packing():
    """A host frame's sample list travels on the frame stream ("samples first") into the list buffer that the frame two
    before it used.  When the host is several untracked frames ahead of the GPU -- here twelve frames in device memory
    integrated at given poses through the queue, 100 us of GPU work each, handed over in 20 us each -- the launch that
    packs that earlier frame (and writes ITS list into the same buffer) may not have run yet when the copy is issued: the
    copy must be ordered behind it, or the tracker reads the older frame's samples.  Pose and volume equal the same
    sequence with a tsdf_synchronize after every frame, bit for bit."""
    import torch
    import tracking_sdf_amd as ts
    w, h, m, n = 640, 480, 512, 14
    seq = synth.Sequence(n_frames=n, width=w, height=h, noise=True, holes=0.02, step=2)
    dev = [seq.frame_torch(k, "cuda") for k in range(n - 1)]
    first = tuple(np.ascontiguousarray(a) for a in seq.frame(0))
    last = tuple(np.ascontiguousarray(a) for a in seq.frame(n - 1))
    torch.cuda.synchronize()

    def run(sync_every_frame):
        s = ts.SDF(m, with_color=False)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        q = lambda k: s.queue_frame_device(dev[k][0].data_ptr(), dev[k][1].data_ptr(), 0, w, h, keep=dev[k])
        # (a host frame first: staging planes, device blocks and sample buffers exist from here on -- allocating them later
        # would synchronise the streams and hide what this test is about)
        t.set_camera_transformation(seq.R[0], seq.t[0])
        s.set_frame(first[0], first[1], None)
        s.update(want_stats=False)
        s.synchronize()
        q(1)
        for k in range(1, n - 1):
            s.next_frame()
            if k + 1 < n - 1:
                q(k + 1)                                  # packed, sample list included, by frame k's integrate launch
            t.set_camera_transformation(seq.R[k], seq.t[k])
            s.update(want_stats=False)
            if sync_every_frame:
                s.synchronize()
        s.set_frame(last[0], last[1], None)               # pageable planes: the sample list goes ahead
        st = t.estimate_new_position()
        pose = (t.rot.copy(), t.trans.copy(), st["iterations"])
        s.update(want_stats=False)
        D, Wt = s.download()
        s.close()
        return pose, D, Wt
    want = run(True)
    for _ in range(3):
        got = run(False)
        assert got[0][2] == want[0][2] and np.array_equal(got[0][0], want[0][0]) and np.array_equal(got[0][1], want[0][1])
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])


def test_the_whole_path_gives_the_same_trajectory_through_the_host_queue():
    """All 1246 poses of the synthetic fr1/plant path at 640x480, 256^3 with colour: frames handed over in device memory
    one at a time against the same frames copied to pageable host memory and handed over through the queue, two frames
    waiting (three host buffers in rotation, as a reader thread would hold them).  1245 tracked poses and the final volume
    equal bit for bit -- the routes differ in when records and sample lists are written, over a long run with every
    buffer of every ring reused hundreds of times."""
    import torch
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=None, width=640, height=480, noise=True, holes=0.02)      # the whole trajectory file
    n = len(seq)
    assert n > 1200
    lib = ts.lib()

    def run(route):
        import ctypes as C
        s = ts.SDF(256, with_color=True)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        poses = np.zeros((n, 12))
        ring = [None] * 4                                   # host copies of the frames in flight (current + two waiting + one being made)

        def to_host(k):
            f = seq.frame_torch(k, "cuda")
            ring[k % 4] = tuple(np.ascontiguousarray(x.cpu().numpy()) for x in f)
            return ring[k % 4]
        if route == "queue":
            s.queue_frame(*to_host(0))
            s.queue_frame(*to_host(1))
        for k in range(n):
            if route == "device":
                f = seq.frame_torch(k, "cuda")
                torch.cuda.synchronize()                   # rendered on torch's stream; the library reads it on its own
                s.set_frame_device(f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(), 640, 480, keep=f)
            else:
                s.next_frame()
                if k + 2 < n:
                    s.queue_frame(*to_host(k + 2))
            s._check(lib.tsdf_track_and_integrate(s._h, 1 if k else 0, None, None))
            poses[k, :9], poses[k, 9:] = t.rot.ravel(), t.trans
        out = (poses, s.download(), s.download_color())
        s.close()
        return out
    want, got = run("device"), run("queue")
    assert np.array_equal(want[0], got[0])
    assert np.linalg.norm(want[0][-1, 9:] - seq.t[n - 1]) < 0.15          # and it is the path, not a stand-still
    for a_, b_ in zip(want[1] + tuple(want[2]), got[1] + tuple(got[2])):
        assert np.array_equal(a_, b_)


@pytest.mark.parametrize("kind", ["pageable", "aos", "depth"])
def test_a_handle_can_be_closed_with_frames_still_waiting(kind):
    """tsdf_destroy with two frames in the queue whose staging may still be running on the library thread: it waits for
    the thread, frees everything, and the caller's buffers are untouched -- twenty times over, no hang, no crash."""
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=3, width=W, height=H, noise=True, holes=0.02, step=4)
    frames = [tuple(np.ascontiguousarray(a) for a in seq.frame(k)) for k in range(3)]
    before = [tuple(a.copy() for a in f) for f in frames]
    z16 = [np.clip(np.where(np.isnan(f[0][..., 2]), 0.0, f[0][..., 2]) * 5000.0, 0, 65535).astype(np.uint16) for f in frames]
    for rep in range(20):
        s = ts.SDF(M, with_color=True)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        def q(k):
            if kind == "aos":
                s.queue_frame_aos(*clouds(*frames[k]))
            elif kind == "depth":
                s.queue_depth_frame(z16[k], frames[k][2], depth_scale=1.0 / 5000.0, sigma_s=3.0, sigma_r=0.05, normal_radius=3)
            else:
                s.queue_frame(*frames[k])
        q(0); q(1)
        if rep % 3 == 1:
            s.next_frame(); s.update(); q(2)
        elif rep % 3 == 2:
            s.synchronize()
        s.close()
    for f, g in zip(frames, before):
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(f, g))


def test_a_queued_frame_of_another_size_is_refused_whatever_waits():
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=2, width=W, height=H, noise=False, step=4)
    xyz, nrm, rgb = (np.ascontiguousarray(a) for a in seq.frame(0))
    s = ts.SDF(M, with_color=True)
    ts.CameraTracking(sdf=s).set_K(seq.K)
    s.queue_frame(xyz, nrm, rgb)                           # no current frame yet: the queued frame sets the size
    big = np.zeros((2 * H, 2 * W, 3), np.float32)
    with pytest.raises(ts.TsdfError):
        s.queue_frame(big, big, np.zeros((2 * H, 2 * W, 3), np.uint8))
    s.queue_frame(xyz, nrm, rgb)
    s.next_frame(); s.update()
    s.next_frame(); s.update()
    with pytest.raises(ts.TsdfError):
        s.next_frame()
    s.close()


def test_update_integrates_the_cloud_as_it_is_when_update_is_called():
    """sdf.cpp:258-259 reads the cloud at update time.  Between estimate_new_position and update the caller may (a) change
    a handful of points of the tracked cloud in place -- the 32-point token of round 4's shim would not have seen them -- or (b) hand
    over another cloud altogether, or (c) call update twice: each time the volume must be what the planar path gives for
    the points update received, and the tracked pose what tracking the ORIGINAL cloud gives."""
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=3, width=W, height=H, noise=True, holes=0.02, step=4)
    fr = [seq.frame(k) for k in range(3)]

    def run(entry_points, variant, whole=False):
        s = ts.SDF(M, with_color=True)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        s.set_frame(*fr[0]); s.update()
        xyz, nrm, rgb = (a.copy() for a in fr[1])
        pts, nn = clouds(xyz, nrm, rgb)
        if entry_points:
            st = s.track_aos(pts, nn if whole else None)
        else:
            s.set_frame(xyz, None, rgb)
            st = t.estimate_new_position()
        pose = (t.rot.copy(), t.trans.copy(), st["iterations"])
        if variant == "one point":              # "a handful": 8 of the 19 200 points (a pixel without a voxel in reach changes nothing)
            valid = np.argwhere(np.isfinite(xyz[..., 2]) & np.isfinite(nrm[..., 2]))
            for r, c in valid[np.linspace(500, len(valid) - 500, 8).astype(int)]:
                xyz[r, c, 2] += 0.05; pts["z"][r, c] = xyz[r, c, 2]
                rgb[r, c, 1] ^= 0x40; pts["g"][r, c] = rgb[r, c, 1]
        elif variant == "one normal":            # ... or a handful of normals (only the whole-frame staging holds them yet)
            valid = np.argwhere(np.isfinite(xyz[..., 2]) & np.isfinite(nrm[..., 2]))
            for r, c in valid[np.linspace(700, len(valid) - 700, 8).astype(int)]:
                nrm[r, c] = nrm[r, c] * np.float32(0.5); nn["normal_x"][r, c], nn["normal_y"][r, c], nn["normal_z"][r, c] = nrm[r, c]
        elif variant == "other cloud":
            xyz, nrm, rgb = (a.copy() for a in fr[2])
            pts, nn = clouds(xyz, nrm, rgb)
        if entry_points:
            s.update_aos(pts, nn)
            if variant == "twice":
                s.update_aos(pts, nn)
        else:
            s.set_frame(xyz, nrm, rgb); s.update()
            if variant == "twice":
                s.update()
        out = (pose, s.download(), s.download_color())
        s.close()
        return out
    for variant in ("same", "one point", "one normal", "other cloud", "twice"):
        want = run(False, variant)
        for whole in (False, True):
            got = run(True, variant, whole)
            assert np.array_equal(want[0][0], got[0][0]) and np.array_equal(want[0][1], got[0][1]) and want[0][2] == got[0][2], (variant, whole)
            for a, b in zip(want[1] + want[2], got[1] + got[2]):
                assert np.array_equal(a, b), (variant, whole)
    # and the one changed point really changes the volume (the test would notice a skipped upload)
    a, b = run(True, "same"), run(True, "one point")
    assert not np.array_equal(a[1][0], b[1][0]) and not np.array_equal(a[2][2], b[2][2])
    a, b = run(True, "same", True), run(True, "one normal", True)
    assert not np.array_equal(a[1][0], b[1][0])             # half-length normals halve the point-to-plane distances


def test_track_aos_state_and_argument_checks():
    import ctypes as C
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=2, width=W, height=H, noise=True, holes=0.02, step=4)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    pts0, nn0 = clouds(*seq.frame(0))
    pts1, nn1 = clouds(*seq.frame(1))
    s.update_aos(pts0, nn0)
    serial = s.frame_serial()
    s.track_aos(pts1)
    assert s.frame_serial() == serial + 1
    with pytest.raises(ts.TsdfError) as ei:
        s.update()                                   # the tracked frame has no normals yet
    assert ei.value.code == ts.E_NO_FRAME
    rot, trans = t.rot.copy(), t.trans.copy()
    s.update_aos(None, nn1)
    assert s.frame_serial() == serial + 1            # the same frame, completed
    lay = ts.AosLayout(8, 0, -1, -1, -1, 12, 0)      # 8-byte stride cannot hold three floats
    assert ts.lib().tsdf_track_aos(s._h, C.c_void_p(pts1.ctypes.data), C.byref(lay), W, H, None) == ts.E_BADARG
    assert ts.lib().tsdf_integrate_aos(s._h, None, None, C.byref(lay), W, H, None) == ts.E_BADARG
    # a failing tracker call (no valid sample) leaves the pose alone and the handle usable
    nanpts = pts1.copy(); nanpts["x"] = np.nan
    with pytest.raises(ts.TsdfError) as ei:
        s.track_aos(nanpts)
    assert ei.value.code == ts.E_NO_SAMPLES and np.array_equal(t.rot, rot) and np.array_equal(t.trans, trans)
    s.track_aos(pts1)
    s.update_aos(pts1, nn1)
    s.close()
