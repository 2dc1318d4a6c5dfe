"""When do the wavefronts of integrate_kernel end their item loops?  (needs a -DTSDF_WG_FINISH=1 build:
tools/build_variants.sh fin="-DTSDF_WG_FINISH=1"; TSDF_HIP_LIB=build/variants/libtsdf_hip_fin.so TSDF_WG_FINISH=1 python tools/wg_finish_probe.py)
Runs the benchmark's frame loop (config 3: 512^3, 640x480, colour) for a few frames and lets tsdf_read_counters print the
distribution of the last launch: loop ends in microseconds after the first wavefront's loop start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tracking_sdf_amd as ts
from tracking_sdf_amd import synth

M, W, H, N = 512, 640, 480, 12
seq = synth.Sequence(n_frames=N, width=W, height=H, noise=True, holes=0.02, step=1)
s = ts.SDF(M, with_color=True)
t = ts.CameraTracking(sdf=s)
t.set_K(seq.K)
for k in range(N):
    s.set_frame(*seq.frame(k))
    if k > 0:
        t.estimate_new_position()
    s.update()
    if k >= N - 3:
        s.read_counters()
s.close()
