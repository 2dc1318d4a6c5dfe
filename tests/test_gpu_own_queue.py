"""Gauss-Newton passes >= 1 through the library's own AQL queue (csrc/aql_queue.cpp).  Opt-in since round 6.

The queue is a submission path, not an algorithm: with it (TSDF_AQL=1, read by tsdf_create; lib/tsdf_track.hsaco lies next
to the library and carries the library's build id) and without it (the default) every pose, every voxel and every pass
count must be the same bits.  The first test also fails when the queue is silently NOT in use although it was asked for."""
import numpy as np
import pytest

from tracking_sdf_amd import synth

pytestmark = pytest.mark.gpu

W, H, M, N = 160, 120, 48, 8


def run(monkeypatch, aql, device_frames):
    import torch
    import tracking_sdf_amd as ts
    if aql is None:
        monkeypatch.delenv("TSDF_AQL", raising=False)
    else:
        monkeypatch.setenv("TSDF_AQL", aql)
    seq = synth.Sequence(n_frames=N, width=W, height=H, noise=True, holes=0.02, step=4)
    s = ts.SDF(M, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    poses, iters, keep = [], [], []
    for k in range(N):
        if device_frames:
            fr = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in seq.frame(k)]
            torch.cuda.synchronize()
            keep.append(fr)
            s.set_frame_device(fr[0].data_ptr(), fr[1].data_ptr(), fr[2].data_ptr(), W, H, keep=fr)
        else:
            s.set_frame(*seq.frame(k))
        if k > 0:
            st = t.estimate_new_position()
            iters.append(st["iterations"] if isinstance(st, dict) else None)
        s.update()
        poses.append((t.rot.copy(), t.trans.copy()))
    cn = s.read_counters()
    D, Wt = s.download()
    col = s.download_color()
    s.close()
    return poses, iters, cn, D, Wt, col


@pytest.mark.parametrize("device_frames", [False, True])
def test_passes_through_the_own_queue_give_the_same_bits_as_through_the_stream(monkeypatch, device_frames):
    on = run(monkeypatch, "1", device_frames)
    off = run(monkeypatch, None, device_frames)
    # the queue really carried the later passes (every pass but the first of each tracked frame) -- and none by default
    assert off[2]["track_passes_own_queue"] == 0
    later = on[2]["track_iterations"] - on[2]["track_calls"]
    assert later > 0 and on[2]["track_passes_own_queue"] == later, on[2]
    assert on[1] == off[1]
    for (r0, t0), (r1, t1) in zip(on[0], off[0]):
        assert np.array_equal(r0, r1) and np.array_equal(t0, t1)
    assert np.array_equal(on[3], off[3]) and np.array_equal(on[4], off[4])
    for a, b in zip(on[5], off[5]):
        assert np.array_equal(a, b)
    for k in ("n_updated", "track_iterations", "track_terms", "track_in_grid", "integrate_items"):
        assert on[2][k] == off[2][k], k


def test_two_handles_keep_their_own_queues_apart(monkeypatch):
    """two trackers alternating frame by frame (two AQL queues, two kernarg rings, one device): each equals a run alone"""
    import tracking_sdf_amd as ts
    alone = run(monkeypatch, "1", False)
    seqs = [synth.Sequence(n_frames=N, width=W, height=H, noise=True, holes=0.02, step=4) for _ in range(2)]
    ss = [ts.SDF(M, with_color=True) for _ in range(2)]
    tt = [ts.CameraTracking(sdf=s) for s in ss]
    for t, q in zip(tt, seqs):
        t.set_K(q.K)
    for k in range(N):
        for s, t, q in zip(ss, tt, seqs):
            s.set_frame(*q.frame(k))
            if k > 0:
                t.estimate_new_position()
            s.update()
    for s, t in zip(ss, tt):
        assert np.array_equal(t.rot, alone[0][-1][0]) and np.array_equal(t.trans, alone[0][-1][1])
        D, Wt = s.download()
        assert np.array_equal(D, alone[3]) and np.array_equal(Wt, alone[4])
        assert s.read_counters()["track_passes_own_queue"] > 0
        s.close()


def test_a_code_object_of_another_build_is_refused(tmp_path):
    """A tsdf_track.hsaco whose build id is not the library's (a stale or foreign file next to the .so) must not run in place
    of the library's kernel: with it, TSDF_AQL=2 says why the queue is not in use, no pass goes through the queue, and the
    results are the stream's."""
    import os
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "tracking_sdf_amd", "lib")
    shutil.copy(os.path.join(libdir, "libtsdf_hip.so"), tmp_path / "libtsdf_hip.so")
    co = open(os.path.join(libdir, "tsdf_track.hsaco"), "rb").read()
    m = re.search(rb"(?<![0-9a-f])[0-9a-f]{32}\x00", co)
    assert m
    bad = bytearray(co)
    bad[m.start()] = ord("f") if co[m.start()] != ord("f") else ord("0")
    (tmp_path / "tsdf_track.hsaco").write_bytes(bytes(bad))
    code = (
        "import numpy as np, tracking_sdf_amd as ts\n"
        "from tracking_sdf_amd import synth\n"
        f"seq = synth.Sequence(n_frames=4, width={W}, height={H}, noise=True, holes=0.02, step=4)\n"
        f"s = ts.SDF({M}, with_color=True); t = ts.CameraTracking(sdf=s); t.set_K(seq.K)\n"
        "for k in range(4):\n"
        "    s.set_frame(*seq.frame(k))\n"
        "    if k: t.estimate_new_position()\n"
        "    s.update()\n"
        "cn = s.read_counters(); print('OWNQ', cn['track_passes_own_queue'], cn['track_iterations'], float(t.trans[0]))\n")
    outs = {}
    for name, libpath in (("bad", str(tmp_path / "libtsdf_hip.so")), ("good", os.path.join(libdir, "libtsdf_hip.so"))):
        env = dict(os.environ, TSDF_AQL="2", TSDF_HIP_LIB=libpath, PYTHONPATH=root)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = (p.stdout.strip().splitlines()[-1].split(), p.stderr)
    assert int(outs["good"][0][1]) > 0                               # the right code object: the queue carries the later passes
    assert int(outs["bad"][0][1]) == 0 and "not this library's build" in outs["bad"][1]
    assert outs["bad"][0][2:] == outs["good"][0][2:]                 # same passes, same pose

