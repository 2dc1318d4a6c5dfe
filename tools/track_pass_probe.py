#!/usr/bin/env python3
"""Wall clock of one tracker accumulation pass (launch, kernel, fan-in, row on the host) in a tight ctypes loop --
steadier than bench.py's per-frame residual (+-0.1 us from run to run): for A/B runs of track_kernel variants
(TSDF_HIP_LIB) and of the fan-in switches (TSDF_HOST_FANIN).  512^3, 640x480, after 8 fused frames; prints one line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth
    n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    dev = torch.device("cuda", 0)
    n = 8
    seq = synth.Sequence(n_frames=n, width=640, height=480, noise=True, holes=0.02, step=8)
    d = [seq.frame_torch(k, dev) for k in range(n)]
    torch.cuda.synchronize()
    s = ts.SDF(512, with_color=True)
    t = ts.CameraTracking(sdf=s)
    t.set_K(seq.K)
    for k in range(n):
        t.set_camera_transformation(seq.R[k], seq.t[k])
        s.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), 640, 480, keep=d[k])
        s.update()
    t.set_camera_transformation(seq.R[n - 1], seq.t[n - 1] + np.array([0.01, -0.01, 0.005]))
    L = ts.lib()
    A, b = np.zeros(36), np.zeros(6)
    pa, pb = A.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double))
    f, h = L.tsdf_accumulate, s._h
    for _ in range(100):
        f(h, pa, pb, None)
    s.synchronize()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n_pass):
            f(h, pa, pb, None)
        reps.append((time.perf_counter() - t0) / n_pass * 1e6)
    print(json.dumps({"track_pass_wall_us": [round(x, 2) for x in reps], "passes_per_repetition": n_pass}))
    s.close()


if __name__ == "__main__":
    main()
