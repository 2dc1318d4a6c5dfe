"""Shared helpers for the parity tests: build the oracle and the HIP path on identical inputs."""
import numpy as np

import oracle as orc
from tracking_sdf_amd import synth

VOL = dict(width=6.0, height=6.0, depth=3.5, origin=(-3.0, -3.0, -0.5), delta=0.3, epsilon=0.025)


def scaled_K(width, height):
    return synth.default_intrinsics(width, height)


def make_oracle(m, K, vol=VOL, gn=(20, 0.001, 1.0, 0.01)):
    s = orc.SDF(m, vol["width"], vol["height"], vol["depth"], vol["origin"], vol["delta"], vol["epsilon"])
    t = orc.CameraTracking(s, *gn)
    t.set_K(K)
    return s, t


def make_gpu(m, K, vol=VOL, gn=(20, 0.001, 1.0, 0.01), **kw):
    import tracking_sdf_amd as ts
    s = ts.SDF(m, vol["width"], vol["height"], vol["depth"], vol["origin"], vol["delta"], vol["epsilon"],
               gn_max_iter=gn[0], max_twist_diff=gn[1], v_h=gn[2], w_h=gn[3], **kw)
    t = ts.CameraTracking(gn[0], gn[1], gn[2], gn[3], s)
    t.set_K(K)
    return s, t


def ulp_diff(a, b):
    """Distance in float32 ulps between two float32 arrays (NaN == NaN counts as 0)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def sym_rel_err(A, B):
    """max |A - B| / max|B| -- scale-relative error for the normal equations."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    return float(np.max(np.abs(A - B)) / max(np.max(np.abs(B)), 1e-300))
