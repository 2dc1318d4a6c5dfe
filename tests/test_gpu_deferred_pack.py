"""Deferred packing of frames handed over in device memory (tsdf_set_frame_device / tsdf_queue_frame_device).

Nothing is launched when such a frame is set: the tracker's first pass reads its samples from the caller's xyz plane
(and leaves them in the sample list), and the pixel records are written inside the integrate launch by workgroups
appended to list_rows_kernel.  Only WHEN the records are written changes -- every pose and every voxel must equal the
host-plane path (tsdf_set_frame: pack_kernel at once) bit for bit, in every order of calls the C ABI allows.
TSDF_DEFER_PACK is read by tsdf_create: 0 = pack when the frame is set, 2 = every pass reads the plane."""
import numpy as np
import pytest

from tracking_sdf_amd import synth

pytestmark = pytest.mark.gpu

W, H, M, N = 160, 120, 48, 5


def frames_on_device(seq, n):
    import torch
    out = []
    for k in range(n):
        out.append([torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in seq.frame(k)])
    torch.cuda.synchronize()
    return out


def finish(s, t, poses):
    D, Wt = s.download()
    col = s.download_color()
    s.close()
    return poses, D, Wt, col


def host_loop(n=N, integrate_every=1):
    """the reference's callback on host planes: pack_kernel runs when the frame is set"""
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    poses, ab = [], []
    for k in range(n):
        s.set_frame(*seq.frame(k))
        if k > 0:
            t.estimate_new_position()
        if k % integrate_every == 0:
            s.update()
        poses.append((t.rot.copy(), t.trans.copy()))
        ab.append(t.accumulate())                 # one more pass at the final pose, after the integration
    return finish(s, t, poses) + (ab,)


def device_loop(n=N, integrate_every=1):
    import tracking_sdf_amd as ts
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    fr = frames_on_device(seq, n)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    poses, ab = [], []
    for k in range(n):
        s.set_frame_device(fr[k][0].data_ptr(), fr[k][1].data_ptr(), fr[k][2].data_ptr(), W, H, keep=fr[k])
        if k > 0:
            t.estimate_new_position()             # first pass: samples from the plane; later passes: from the list it wrote
        if k % integrate_every == 0:
            s.update()                            # records packed inside this launch
        poses.append((t.rot.copy(), t.trans.copy()))
        ab.append(t.accumulate())                 # k == 0: the list the packing workgroups wrote (no tracker pass before)
    return finish(s, t, poses) + (ab,)


def assert_same(want, got):
    for (r0, t0), (r1, t1) in zip(want[0], got[0]):
        assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
    assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])
    for a, b in zip(want[3], got[3]):
        assert np.array_equal(a, b)
    if len(want) > 4:
        for (A0, b0, st0), (A1, b1, st1) in zip(want[4], got[4]):
            assert np.array_equal(A0, A1) and np.array_equal(b0, b1) and st0 == st1


@pytest.mark.parametrize("mode", [None, "0", "2"])
def test_device_frames_equal_host_frames_bit_for_bit(mode, monkeypatch):
    if mode is None:
        monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    else:
        monkeypatch.setenv("TSDF_DEFER_PACK", mode)
    assert_same(host_loop(), device_loop())


def test_frames_that_are_tracked_but_not_integrated(monkeypatch):
    """a deferred packing that no integrate launch picks up is dropped with the frame; the next frame starts clean"""
    monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    assert_same(host_loop(integrate_every=2), device_loop(integrate_every=2))


def test_queued_device_frames_in_every_order(monkeypatch):
    """tsdf_queue_frame_device: packed by the CURRENT frame's integrate launch when one comes by (sample list included),
    otherwise taken over unpacked by tsdf_next_frame like a frame of tsdf_set_frame_device"""
    import tracking_sdf_amd as ts
    monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    n = 6
    want = host_loop(n=n, integrate_every=2)
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    fr = frames_on_device(seq, n)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    poses, ab = [], []

    def queue(k):
        s.queue_frame_device(fr[k][0].data_ptr(), fr[k][1].data_ptr(), fr[k][2].data_ptr(), W, H, keep=fr[k])
    queue(0)
    for k in range(n):
        s.next_frame()
        if k + 1 < n and k % 3 != 2:
            queue(k + 1)                          # before the hot calls: this frame's integrate launch (if any) packs it
        if k > 0:
            t.estimate_new_position()
        if k % 2 == 0:
            s.update()
        if k + 1 < n and k % 3 == 2:
            queue(k + 1)                          # after them: stays unpacked until it is current
        poses.append((t.rot.copy(), t.trans.copy()))
        ab.append(t.accumulate())
    assert_same(want, finish(s, t, poses) + (ab,))


def test_synchronize_ends_the_claim_on_the_planes(monkeypatch):
    """tsdf_synchronize packs what is still deferred: afterwards the caller may overwrite or free its device planes --
    also between a frame's tracking and its integration, and for a queued frame that no integrate launch has seen"""
    import torch
    import tracking_sdf_amd as ts
    monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    want = host_loop(n=4)
    seq = synth.Sequence(n_frames=4, width=W, height=H, noise=True, holes=0.02, step=4)
    fr = frames_on_device(seq, 4)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    poses, ab = [], []

    def scratch(k):
        buf = [a.clone() for a in fr[k]]
        return buf

    for k in range(4):
        buf = scratch(k)
        if k == 2:                                   # through the queue, overwritten while it is only queued
            s.queue_frame_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), W, H, keep=buf)
            s.synchronize()
            for a in buf:
                a.zero_()
            torch.cuda.synchronize()
            s.next_frame()
        else:
            s.set_frame_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), W, H, keep=buf)
        if k > 0:
            t.estimate_new_position()
        if k != 2:
            s.synchronize()                          # k == 0: nothing read the planes yet; otherwise only the tracker did
            for a in buf:
                a.zero_()
            torch.cuda.synchronize()
        s.update()
        poses.append((t.rot.copy(), t.trans.copy()))
        ab.append(t.accumulate())
    assert_same(want, finish(s, t, poses) + (ab,))


@pytest.mark.parametrize("w,h,stride,color", [(203, 117, 3, True), (64, 48, 1, False), (161, 120, 4, True), (97, 33, 5, False)])
def test_plane_sampling_for_odd_sizes_and_strides(w, h, stride, color, monkeypatch):
    """the tracker's plane reads index pixel (col * stride, row * stride) of a `width`-wide image in the reference's visiting
    order (columns outer); image sizes that are no multiples of the stride or of the packing tiles, with and without
    colour records, against the host-plane loop bit for bit"""
    import tracking_sdf_amd as ts
    monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    n = 3
    seq = synth.Sequence(n_frames=n, width=w, height=h, noise=True, holes=0.02, step=4)

    def run(device):
        s = ts.SDF(40, with_color=color, pixel_stride=stride)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        fr = frames_on_device(seq, n) if device else None
        out = []
        for k in range(n):
            if device:
                s.set_frame_device(fr[k][0].data_ptr(), fr[k][1].data_ptr(), fr[k][2].data_ptr() if color else 0, w, h, keep=fr[k])
            else:
                xyz, nrm, rgb = seq.frame(k)
                s.set_frame(xyz, nrm, rgb if color else None)
            if k > 0:
                st = t.estimate_new_position()
                out.append((t.rot.copy(), t.trans.copy(), st["iterations"]))
            out.append(t.accumulate()[:2])
            s.update()
        D, Wt = s.download()
        s.close()
        return out, D, Wt
    a, b = run(False), run(True)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for x, y in zip(a[0], b[0]):
        for u, v in zip(x, y):
            assert np.array_equal(np.asarray(u), np.asarray(v))


@pytest.mark.parametrize("ring,mode", [(3, None), (2, None), (1, None), (2, "0"), (1, "0"), (3, "queue"), (2, "queue")])
def test_ring_of_device_buffers_refilled_as_soon_as_they_are_reported_free(ring, mode, monkeypatch):
    """The borrowing rule of ABI version 3 (tsdf.h, tsdf_set_frame_device): a frame's device planes are the library's until
    tsdf_device_frame_released() reaches its serial.  A producer with a ring of `ring` device buffers overwrites a buffer
    with NaN THE MOMENT the call reports it free -- on torch's stream, which is not ordered against the library's, so a
    report that came too early would put NaN into the pixel records -- and refills it with a later frame only then.  With
    three buffers the loop never has to wait; with ONE buffer it polls behind every integrate launch (the report comes a few
    microseconds into the launch that packs the frame, while integrate_kernel is still running): either way 40 frames give the trajectory and the volume of the host-plane loop, bit
    for bit.  mode "0": TSDF_DEFER_PACK=0 (packed by a launch of its own when set); "queue": through
    tsdf_queue_frame_device (packed inside the PREVIOUS frame's integrate launch)."""
    import time
    import torch
    import tracking_sdf_amd as ts
    if mode == "0":
        monkeypatch.setenv("TSDF_DEFER_PACK", "0")
    else:
        monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    n = 40
    want = host_loop(n=n)[:4]
    seq = synth.Sequence(n_frames=n, width=W, height=H, noise=True, holes=0.02, step=4)
    src = frames_on_device(seq, n)                     # the producer's source; the library only ever sees the ring
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    bufs = [[torch.empty_like(a) for a in src[0]] for _ in range(ring)]
    holds = [None] * ring                              # serial of the frame a buffer holds (None: free)
    last_rel = [s.device_frame_released()]
    waits = 0

    def reclaim():
        rel = s.device_frame_released()
        assert rel >= last_rel[0] and rel <= s.frame_serial() + 1         # monotonic, never ahead of what was handed over
        last_rel[0] = rel
        for b in range(ring):
            if holds[b] is not None and holds[b] <= rel:
                bufs[b][0].fill_(float("nan")); bufs[b][1].fill_(float("nan")); bufs[b][2].zero_()   # torch's stream: at once
                holds[b] = None
        return rel

    def produce(k, serial):
        nonlocal waits
        t0 = time.perf_counter()
        while True:
            reclaim()
            free = [b for b in range(ring) if holds[b] is None]
            if free:
                break
            waits += 1
            assert time.perf_counter() - t0 < 5.0, "no buffer was ever reported free"
        b = free[0]
        for dst, a in zip(bufs[b], src[k]):
            dst.copy_(a)
        torch.cuda.synchronize()                       # "contents must be complete when the call is made"
        holds[b] = serial
        return bufs[b]

    poses = []
    if mode == "queue":
        buf = produce(0, 1)
        s.queue_frame_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), W, H)
    for k in range(n):
        if mode == "queue":
            s.next_frame()
            assert s.frame_serial() == k + 1
            if k + 1 < n:
                buf = produce(k + 1, k + 2)
                s.queue_frame_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), W, H)
        else:
            buf = produce(k, k + 1)
            s.set_frame_device(buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), W, H)
            assert s.frame_serial() == k + 1
        if k > 0:
            t.estimate_new_position()
        s.update(want_stats=False)
        reclaim()                                      # right behind the integrate launch: the frame may or may not be free yet
        poses.append((t.rot.copy(), t.trans.copy()))
    s.synchronize()
    assert s.device_frame_released() == s.frame_serial() == n
    reclaim()
    assert all(x is None for x in holds)
    got = finish(s, t, poses)
    assert_same(want, got)
    if ring == 3 and mode is None:
        assert waits == 0                              # three buffers never wait in the plain loop


def test_a_frame_that_is_replaced_unpacked_is_free_at_once(monkeypatch):
    """set_frame_device(A), track, set_frame_device(B) without integrating A: nothing will ever pack A (its tracker passes
    were host-synchronous), so A is reported free as soon as B is set; B stays borrowed until its integrate launch has run."""
    import time
    import tracking_sdf_amd as ts
    monkeypatch.delenv("TSDF_DEFER_PACK", raising=False)
    seq = synth.Sequence(n_frames=3, width=W, height=H, noise=True, holes=0.02, step=4)
    fr = frames_on_device(seq, 3)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    assert s.device_frame_released() == 0
    s.set_frame_device(fr[0][0].data_ptr(), fr[0][1].data_ptr(), fr[0][2].data_ptr(), W, H)
    assert s.device_frame_released() == 0              # handed over, not packed
    s.update()                                         # with statistics: synchronous
    assert s.device_frame_released() == 1
    s.set_frame_device(fr[1][0].data_ptr(), fr[1][1].data_ptr(), fr[1][2].data_ptr(), W, H)
    t.estimate_new_position()
    assert s.device_frame_released() == 1              # frame 2 only tracked so far
    s.set_frame_device(fr[2][0].data_ptr(), fr[2][1].data_ptr(), fr[2][2].data_ptr(), W, H)
    assert s.device_frame_released() == 2              # frame 2 was replaced unpacked: free
    s.set_frame(*seq.frame(2))                         # a host frame replaces device frame 3 (serial 4): 3 is free, 4 never was borrowed
    assert s.device_frame_released() == 4 == s.frame_serial()
    s.close()
