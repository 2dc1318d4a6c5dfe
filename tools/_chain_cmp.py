import os, sys, numpy as np
sys.path.insert(0, '.')
import tracking_sdf_amd as ts
from tracking_sdf_amd import synth
W, H, M = 160, 120, 48
seq = synth.Sequence(n_frames=3, width=W, height=H, noise=True, holes=0.02, step=4)
def run(chain, k):
    os.environ['TSDF_TRACK_CHAIN'] = '1' if chain else '0'
    s = ts.SDF(M, with_color=True); t = ts.CameraTracking(gauss_newton_max_iteration=k, sdf=s); t.set_K(seq.K)
    s.set_frame(*seq.frame(0)); s.update()
    s.set_frame(*seq.frame(1))
    st = t.estimate_new_position()
    out = (t.rot.copy(), t.trans.copy(), st)
    s.close()
    return out
for k in (1, 2, 3, 4, 6, 20):
    a = run(False, k); b = run(True, k)
    print(k, 'iters', a[2]['iterations'], b[2]['iterations'], 'stopped', a[2]['stopped'], b[2]['stopped'],
          'drot %.3e dtrans %.3e' % (np.abs(a[0]-b[0]).max(), np.abs(a[1]-b[1]).max()), 'twist', np.abs(np.array(a[2]['last_twist'])-np.array(b[2]['last_twist'])).max())
