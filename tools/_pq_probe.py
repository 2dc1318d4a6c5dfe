import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import tracking_sdf_amd as ts
from tracking_sdf_amd import synth
dev = torch.device('cuda', 0)
n = 8
seq = synth.Sequence(n_frames=n, width=640, height=480, noise=True, holes=0.02, step=8)
d = [seq.frame_torch(k, dev) for k in range(n)]
torch.cuda.synchronize()
s = ts.SDF(512, with_color=True); t = ts.CameraTracking(sdf=s); t.set_K(seq.K)
for k in range(n):
    t.set_camera_transformation(seq.R[k], seq.t[k])
    s.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), 640, 480, keep=d[k]); s.update()
t.set_camera_transformation(seq.R[n-1], seq.t[n-1] + np.array([0.01, -0.01, 0.005]))
L = ts.lib(); A = np.zeros(36); b = np.zeros(6)
pa, pb = A.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double))
f = L.tsdf_accumulate; h = s._h
for _ in range(50): f(h, pa, pb, None)
s.synchronize()
N = 4000
t0 = time.perf_counter()
for _ in range(N): f(h, pa, pb, None)
dt = (time.perf_counter() - t0) / N
print('pass wall us %.2f' % (dt * 1e6))
