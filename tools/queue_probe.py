#!/usr/bin/env python3
"""Pageable frames through the library's frame queue, nothing else: frames/s for --ahead frames waiting, with the staging
profile (TSDF_PROFILE=1 prints it when the volume is closed).  python3 tools/queue_probe.py --ahead 2 --frames 60"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tracking_sdf_amd as ts
from tracking_sdf_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--ahead", type=int, default=2)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--warmup", type=int, default=6)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--kind", default="pageable", choices=["pageable", "pinned", "device", "device_set", "set", "set_aos", "ref"])
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--times", action="store_true", help="wall time inside the frame call and inside the hot call, per frame")
a = ap.parse_args()
n = a.frames + a.warmup + 1
seq = synth.Sequence(n_frames=n, width=640, height=480, noise=True, holes=0.02)
dev = torch.device("cuda:0")
fr = [seq.frame_torch(k, dev) for k in range(n)]
host = [tuple(np.ascontiguousarray(t.cpu().numpy()) for t in f) for f in fr]
if a.kind == "pinned":
    hold = [tuple(torch.from_numpy(x).pin_memory() for x in f) for f in host]
    host = [tuple(t.numpy() for t in f) for f in hold]
aos = None
if a.kind in ("set_aos", "ref"):
    def clouds(xyz, nrm, rgb):
        pts = np.zeros(xyz.shape[:2], dtype=np.dtype({"names": ["x", "y", "z", "b", "g", "r"], "formats": ["<f4"] * 3 + ["u1"] * 3,
                                                     "offsets": [0, 4, 8, 16, 17, 18], "itemsize": 32}))
        nn = np.zeros(xyz.shape[:2], dtype=np.dtype({"names": ["normal_x", "normal_y", "normal_z"], "formats": ["<f4"] * 3,
                                                    "offsets": [0, 4, 8], "itemsize": 32}))
        pts["x"], pts["y"], pts["z"] = xyz[..., 0], xyz[..., 1], xyz[..., 2]
        pts["r"], pts["g"], pts["b"] = rgb[..., 0], rgb[..., 1], rgb[..., 2]
        nn["normal_x"], nn["normal_y"], nn["normal_z"] = nrm[..., 0], nrm[..., 1], nrm[..., 2]
        return pts, nn
    aos = [clouds(*f) for f in host]
s = ts.SDF(a.m, with_color=True)
t = ts.CameraTracking(sdf=s)
t.set_K(seq.K)
L = ts.lib()
rates = []
for rep in range(a.repeat):
    s.reset()
    def q(i):
        if a.kind == "device":
            s.queue_frame_device(fr[i][0].data_ptr(), fr[i][1].data_ptr(), fr[i][2].data_ptr(), 640, 480, keep=fr[i])
        else:
            s.queue_frame(*host[i])
    ahead = 1 if a.kind == "device" else 0 if a.kind.startswith("set") or a.kind in ("ref", "device_set") else a.ahead
    for j in range(ahead):
        q(j)
    t0 = None
    t_feed = t_hot = 0.0
    for k in range(n):
        if k == a.warmup + 1:
            s.synchronize(); t0 = time.perf_counter(); t_feed = t_hot = 0.0
        ta = time.perf_counter()
        if a.kind == "ref":                       # the reference's two calls (estimate_new_position, update), synchronously
            pass
        elif a.kind == "device_set":              # frames resident in HBM, one at a time
            s.set_frame_device(fr[k][0].data_ptr(), fr[k][1].data_ptr(), fr[k][2].data_ptr(), 640, 480, keep=fr[k])
        elif a.kind == "set":                     # one frame at a time, pageable planes
            s.set_frame(*host[k])
        elif a.kind == "set_aos":
            s.set_frame_aos(*aos[k])
        else:
            s.next_frame()
            if k + ahead < n:
                q(k + ahead)
        tb = time.perf_counter()
        if a.kind == "ref":
            if k > 0:
                s.track_aos(aos[k][0], None)
            tb = time.perf_counter()
            s.update_aos(aos[k][0], aos[k][1])
        elif k == 0:
            s.update(want_stats=False)
        else:
            s._check(L.tsdf_track_and_integrate(s._h, 1, None, None))
        tc = time.perf_counter()
        t_feed += tb - ta; t_hot += tc - tb
    s.synchronize()
    rates.append(a.frames / (time.perf_counter() - t0))
    if a.times:
        print("  us per frame: frame call(s) %.1f  tsdf_track_and_integrate %.1f  whole %.1f" % (t_feed / a.frames * 1e6, t_hot / a.frames * 1e6, 1e6 / rates[-1]), flush=True)
print("queue_probe", a.kind, "ahead", a.ahead, "threads", os.environ.get("TSDF_HOST_THREADS", "default"), "frames/s", [round(r, 1) for r in rates], flush=True)
s.close()
