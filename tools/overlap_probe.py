#!/usr/bin/env python3
"""Does an integrate launch hide under tracker passes?  Two handles (own streams) on one GPU: A runs tracker
accumulation passes (host-polled, like tsdf_track), B runs integrate launches; each is timed alone and together."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tracking_sdf_amd as ts
from tracking_sdf_amd import synth

dev = torch.device("cuda", 0)
seq = synth.Sequence(n_frames=6, width=640, height=480, noise=True, holes=0.02, step=8)
fr = [seq.frame_torch(k, dev) for k in range(6)]
torch.cuda.synchronize()

def make():
    sdf = ts.SDF(512, with_color=True)
    trk = ts.CameraTracking(sdf=sdf)
    trk.set_K(seq.K)
    for k in range(6):
        trk.set_camera_transformation(seq.R[k], seq.t[k])
        sdf.set_frame_device(fr[k][0].data_ptr(), fr[k][1].data_ptr(), fr[k][2].data_ptr(), 640, 480)
        sdf.update(want_stats=False)
    sdf.read_counters()
    return sdf, trk

A = make(); B = make()
A[1].set_camera_transformation(seq.R[5], seq.t[5] + np.array([0.01, -0.01, 0.005]))

def run_track(n, out):
    t0 = time.perf_counter()
    for _ in range(n):
        A[1].accumulate()
    out["track_s"] = time.perf_counter() - t0

def run_integ(n, out):
    t0 = time.perf_counter()
    for _ in range(n):
        B[0].update(want_stats=False)
    B[0].read_counters()          # drains the stream
    out["integ_s"] = time.perf_counter() - t0

NT, NI = 2000, 400
r = {}
run_track(200, {}); run_integ(50, {})
o = {}; run_track(NT, o); r["track_alone_us_per_pass"] = o["track_s"] / NT * 1e6
o = {}; run_integ(NI, o); r["integ_alone_us_per_launch"] = o["integ_s"] / NI * 1e6
o = {}
t1 = threading.Thread(target=run_track, args=(NT, o)); t2 = threading.Thread(target=run_integ, args=(NI, o))
t0 = time.perf_counter(); t1.start(); t2.start(); t1.join(); t2.join(); wall = time.perf_counter() - t0
r["together_track_us_per_pass"] = o["track_s"] / NT * 1e6
r["together_integ_us_per_launch"] = o["integ_s"] / NI * 1e6
r["together_wall_s"] = wall
r["sum_alone_s"] = (r["track_alone_us_per_pass"] * NT + r["integ_alone_us_per_launch"] * NI) * 1e-6
print(json.dumps(r))
