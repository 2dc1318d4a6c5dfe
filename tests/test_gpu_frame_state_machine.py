"""The frame-input state machine under a random walk.

Every way a frame can reach the library -- host planes set one by one, PCL-style clouds, device planes (packed
inside their integrate launch, samples read from the plane by the first tracker pass), and the same three through the
frame queue (queued before or after the current frame's hot calls, page-locked or pageable, one or two frames ahead) -- mixed at random from
frame to frame, with frames that are only tracked, only integrated or neither, extra accumulation passes and
tsdf_synchronize calls thrown in.  Whatever the route, poses, normal equations and the volume must equal the plain
host-plane loop bit for bit: the routes differ in WHEN records and sample lists are written, never in what they hold."""
import numpy as np
import pytest

from tracking_sdf_amd import synth

pytestmark = pytest.mark.gpu

W, H, M, N = 96, 72, 32, 28


def clouds(ts, xyz, nrm, rgb):
    pts = np.zeros(xyz.shape[:2], dtype=ts.PCL_POINT_XYZRGB)
    pts["x"], pts["y"], pts["z"] = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    pts["r"], pts["g"], pts["b"] = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    nn = np.zeros(xyz.shape[:2], dtype=ts.PCL_NORMAL)
    nn["normal_x"], nn["normal_y"], nn["normal_z"] = nrm[..., 0], nrm[..., 1], nrm[..., 2]
    return pts, nn


def plan(seed):
    """per frame: how it arrives, whether it is tracked / integrated, and the extras"""
    rng = np.random.default_rng(seed)
    kinds = ["set_host", "set_aos", "set_device", "q_host", "q_pinned", "q_aos", "q_device"]
    out = []
    for k in range(N):
        out.append({"kind": kinds[rng.integers(len(kinds))] if k > 0 else kinds[rng.integers(3)],
                    "queue_early": bool(rng.integers(2)),          # queued before (True) or after the previous frame's hot calls
                    "two_ahead": bool(rng.integers(2)),            # ... and the frame after it as well, if that one comes from host memory through the queue
                    "track": k > 0 and rng.random() < 0.85, "integrate": k == 0 or rng.random() < 0.8,
                    "accumulate": rng.random() < 0.4, "sync": rng.random() < 0.25})
    return out


def reference(seq, steps):
    import tracking_sdf_amd as ts
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    log = []
    for k, st in enumerate(steps):
        s.set_frame(*seq.frame(k))
        if st["track"]:
            track(ts, t, log)
        if st["accumulate"]:
            log.append(t.accumulate()[:2])
        if st["integrate"]:
            s.update()
        log.append((t.rot.copy(), t.trans.copy()))
    out = (log, s.download(), s.download_color())
    s.close()
    return out


def track(ts, t, log):
    try:
        t.estimate_new_position()
    except ts.TsdfError as e:                              # (a refused pass leaves the pose as it was: part of what is compared)
        log.append((np.array([e.code]),))


def mixed(seq, steps):
    import torch
    import tracking_sdf_amd as ts
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    host = [tuple(np.ascontiguousarray(a) for a in seq.frame(k)) for k in range(N)]
    pinned = [[torch.from_numpy(a.copy()).pin_memory() for a in f] for f in host]
    dev = [[torch.from_numpy(a).cuda() for a in f] for f in host]
    aos = [clouds(ts, *f) for f in host]
    torch.cuda.synchronize()

    def queue(k):
        kind = steps[k]["kind"]
        if kind == "q_host":
            s.queue_frame(*host[k])
        elif kind == "q_pinned":
            s.queue_frame(*[x.numpy() for x in pinned[k]])
        elif kind == "q_aos":
            s.queue_frame_aos(*aos[k])
        else:
            s.queue_frame_device(dev[k][0].data_ptr(), dev[k][1].data_ptr(), dev[k][2].data_ptr(), W, H, keep=dev[k])

    log = []
    queued = set()

    def queue_ahead(k):
        """frame k+1 into the queue, and frame k+2 behind it when the plan says so and the library takes it there (a host
        frame; a frame in device memory waits in the first place only)"""
        if k + 1 not in queued:
            queue(k + 1); queued.add(k + 1)
        if (steps[k + 1]["two_ahead"] and k + 2 < N and steps[k + 2]["kind"] in ("q_host", "q_pinned", "q_aos") and k + 2 not in queued):
            queue(k + 2); queued.add(k + 2)

    for k, st in enumerate(steps):
        kind = st["kind"]
        if kind.startswith("q_"):
            s.next_frame()                                   # queued during frame k-1 (or k-2)
        elif kind == "set_host":
            s.set_frame(*host[k])
        elif kind == "set_aos":
            s.set_frame_aos(*aos[k])
        else:
            s.set_frame_device(dev[k][0].data_ptr(), dev[k][1].data_ptr(), dev[k][2].data_ptr(), W, H, keep=dev[k])
        nxt = steps[k + 1] if k + 1 < N else None
        if nxt and nxt["kind"].startswith("q_") and nxt["queue_early"]:
            queue_ahead(k)
        if st["track"]:
            track(ts, t, log)
        if st["sync"]:
            s.synchronize()
        if st["accumulate"]:
            log.append(t.accumulate()[:2])
        if st["integrate"]:
            s.update()
        if nxt and nxt["kind"].startswith("q_") and not nxt["queue_early"]:
            queue_ahead(k)
        log.append((t.rot.copy(), t.trans.copy()))
    out = (log, s.download(), s.download_color())
    s.close()
    return out


@pytest.mark.parametrize("seed,defer", [(k, None) for k in range(1, 17)] + [(1, "0"), (2, "0"), (3, "2"), (4, "2")])
def test_every_route_of_a_frame_gives_the_same_bits(seed, defer, monkeypatch):
    if defer is None:
        monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    else:
        monkeypatch.setenv("TSDF_DEFER_PACK", defer)      # 0: device frames packed when they are set; 2: every pass reads the plane
    seq = synth.Sequence(n_frames=N, width=W, height=H, noise=True, holes=0.02, step=3)
    steps = plan(seed)
    want, got = reference(seq, steps), mixed(seq, steps)
    assert len(want[0]) == len(got[0])
    for a, b in zip(want[0], got[0]):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    for a, b in zip(want[1] + want[2], got[1] + got[2]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("seed", [1, 2])
def test_every_route_at_full_image_size_with_the_host_ahead_of_the_gpu(seed, monkeypatch):
    """The same walk where timing matters: 640x480 frames into a 256^3 volume, most frames integrated at the pose they arrive
    with (no tracking: nothing makes the host wait, so it runs frames ahead of the GPU -- the conditions under which the two
    cross-stream ordering bugs of round 6 showed), a few synchronised or tracked.  Same bits as the plain loop."""
    import test_gpu_frame_state_machine as me
    monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    for name, value in (("W", 640), ("H", 480), ("M", 256), ("N", 18)):
        monkeypatch.setattr(me, name, value)
    rng = np.random.default_rng(1000 + seed)
    steps = plan(seed)
    for k, st in enumerate(steps):
        st["track"] = k > 0 and rng.random() < 0.35
        st["integrate"] = True
        st["sync"] = rng.random() < 0.1
        st["accumulate"] = rng.random() < 0.15
    seq = synth.Sequence(n_frames=me.N, width=me.W, height=me.H, noise=True, holes=0.02, step=3)
    import functools
    seq.frame = functools.lru_cache(maxsize=None)(seq.frame)          # (rendered on the CPU: once per frame, not once per route)
    want, got = reference(seq, steps), mixed(seq, steps)
    assert len(want[0]) == len(got[0])
    for a, b in zip(want[0], got[0]):
        for x, y in zip(a, b):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    for a, b in zip(want[1] + want[2], got[1] + got[2]):
        assert np.array_equal(a, b)
