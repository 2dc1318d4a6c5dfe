"""tracking_sdf_amd -- MI355X-native hot path of mees/tracking_sdf behind its own entry points.

This package is a thin ctypes binding of the C ABI in include/tsdf.h (libtsdf_hip.so: hand-written
gfx950 HIP kernels + the C++ host loop) plus Python mirrors of the reference's two classes,
``SDF`` (include/sdf_3d_reconstruction/sdf.h) and ``CameraTracking`` (camera_tracking.h), with the
reference's method names.  There is no CPU fallback: if the shared library is missing or no GPU is
visible, the compute entry points raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

__all__ = ["SDF", "CameraTracking", "TsdfError", "Config", "lib", "lib_path", "build", "slab_range", "slab_range_weighted",
           "frustum_layer_weights", "halo_for"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
# TSDF_HIP_LIB: another build of the same library (tuning experiments: tools/build_variants.sh); default = the in-tree one
_LIB_PATH = os.environ.get("TSDF_HIP_LIB") or os.path.join(_HERE, "lib", "libtsdf_hip.so")

# status codes of include/tsdf.h
OK, E_BADARG, E_NO_DEVICE, E_HIP, E_NO_INTRINSICS, E_NO_FRAME, E_SINGULAR, E_NO_SAMPLES, E_HALO, E_COMM, E_NOMEM = \
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10
RED_WIDTH = 34
RED_ALLREDUCE = 30
ABI_VERSION = 4          # TSDF_ABI_VERSION of include/tsdf.h this module mirrors (struct layouts)


class TsdfError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"tsdf error {code}: {message}")
        self.code = code


class Config(C.Structure):
    """struct tsdf_config"""
    _fields_ = [("m", C.c_int32), ("width", C.c_float), ("height", C.c_float), ("depth", C.c_float),
                ("origin", C.c_double * 3), ("delta", C.c_float), ("epsilon", C.c_float),
                ("gn_max_iter", C.c_int32), ("max_twist_diff", C.c_float), ("v_h", C.c_float), ("w_h", C.c_float),
                ("pixel_stride", C.c_int32), ("stale_carry", C.c_int32), ("carry_threads", C.c_int32), ("with_color", C.c_int32),
                ("slab_x0", C.c_int32), ("slab_x1", C.c_int32), ("halo", C.c_int32), ("device", C.c_int32),
                ("slab_stride", C.c_int32)]


class PreprocParams(C.Structure):
    """struct tsdf_preproc_params"""
    _fields_ = [("depth_scale", C.c_float), ("sigma_s", C.c_float), ("sigma_r", C.c_float), ("radius", C.c_int32),
                ("normal_radius", C.c_int32), ("max_depth_change", C.c_float), ("grid_filter", C.c_int32)]


class AosLayout(C.Structure):
    """struct tsdf_aos_layout"""
    _fields_ = [("point_stride", C.c_int32), ("xyz_offset", C.c_int32), ("r_offset", C.c_int32), ("g_offset", C.c_int32),
                ("b_offset", C.c_int32), ("normal_stride", C.c_int32), ("normal_offset", C.c_int32)]


# numpy views of PCL's two point types as the reference holds them (32 bytes each, EIGEN_ALIGN16 unions):
# PointXYZRGB = {x, y, z, pad, b, g, r, a, pad[12]}, Normal = {normal_x, normal_y, normal_z, pad, curvature, pad[12]}
PCL_POINT_XYZRGB = np.dtype({"names": ["x", "y", "z", "b", "g", "r", "a"], "formats": ["<f4", "<f4", "<f4", "u1", "u1", "u1", "u1"],
                             "offsets": [0, 4, 8, 16, 17, 18, 19], "itemsize": 32})
PCL_NORMAL = np.dtype({"names": ["normal_x", "normal_y", "normal_z", "curvature"], "formats": ["<f4"] * 4,
                       "offsets": [0, 4, 8, 16], "itemsize": 32})


class IntegrateStats(C.Structure):
    _fields_ = [("n_updated", C.c_int64), ("n_updated_halo", C.c_int64), ("n_voxels", C.c_int64)]


class AccumStats(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("n_nan", C.c_int64), ("n_oog", C.c_int64),
                ("n_in_grid_owned", C.c_int64), ("n_ok", C.c_int64), ("n_terms", C.c_int64)]


class TrackStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("stopped", C.c_int32), ("n_terms_last", C.c_int64),
                ("last_twist", C.c_double * 6)]


class Timing(C.Structure):
    _fields_ = [("integrate_ms", C.c_double), ("integrate_launches", C.c_int64),
                ("track_ms", C.c_double), ("track_launches", C.c_int64),
                ("pack_ms", C.c_double), ("pack_launches", C.c_int64)]


class Counters(C.Structure):
    _fields_ = [("n_updated", C.c_int64), ("n_updated_halo", C.c_int64), ("n_voxels_swept", C.c_int64),
                ("integrate_calls", C.c_int64), ("track_calls", C.c_int64), ("track_iterations", C.c_int64),
                ("track_in_grid", C.c_int64), ("track_terms", C.c_int64), ("integrate_items", C.c_int64),
                ("track_passes_own_queue", C.c_int64)]


def _struct_dict(s):
    out = {}
    for k, _ in s._fields_:
        v = getattr(s, k)
        out[k] = list(v) if hasattr(v, "__len__") else v
    return out


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int32, C.c_void_p)

# every symbol include/tsdf.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = (
    "tsdf_abi_version", "tsdf_default_config", "tsdf_create", "tsdf_destroy", "tsdf_last_error", "tsdf_strerror",
    "tsdf_get_config", "tsdf_set_intrinsics", "tsdf_set_camera_transformation", "tsdf_set_tracker_params", "tsdf_get_pose", "tsdf_set_frame",
    "tsdf_frame_serial", "tsdf_queue_frame", "tsdf_queue_frame_device", "tsdf_queue_frame_aos", "tsdf_next_frame",
    "tsdf_set_frame_device", "tsdf_device_frame_released", "tsdf_set_frame_aos", "tsdf_track_aos", "tsdf_track_frame_aos", "tsdf_integrate_aos", "tsdf_default_preproc", "tsdf_set_depth_frame", "tsdf_queue_depth_frame", "tsdf_get_preprocessed", "tsdf_integrate", "tsdf_track", "tsdf_track_and_integrate", "tsdf_accumulate", "tsdf_gn_update", "tsdf_sample",
    "tsdf_download", "tsdf_upload", "tsdf_download_color", "tsdf_upload_color", "tsdf_upload_with_halo", "tsdf_upload_color_with_halo", "tsdf_reset", "tsdf_save", "tsdf_load",
    "tsdf_mesh_extract", "tsdf_mesh_read", "tsdf_mesh_device",
    "tsdf_slab_range", "tsdf_cyclic_range", "tsdf_slab_range_weighted", "tsdf_frustum_layer_weights", "tsdf_halo_for", "tsdf_comm_unique_id", "tsdf_comm_init", "tsdf_comm_init_shm", "tsdf_comm_init_peer", "tsdf_comm_finalize", "tsdf_set_allreduce_hook",
    "tsdf_allreduce", "tsdf_host_set_pose", "tsdf_host_perturbed_rotations", "tsdf_host_gn_step", "tsdf_set_timing", "tsdf_read_timing", "tsdf_read_counters", "tsdf_synchronize", "tsdf_stream",
)

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def build(force: bool = False) -> str:
    """Compile libtsdf_hip.so for gfx950 with hipcc (works without a GPU)."""
    import subprocess
    if force and os.path.exists(_LIB_PATH):
        os.remove(_LIB_PATH)
    subprocess.check_call(["make", "-C", _ROOT, "-s", _LIB_PATH])
    return _LIB_PATH


def lib():
    """Load libtsdf_hip.so.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise TsdfError(E_NO_DEVICE, f"{_LIB_PATH} not built: run `make` (hipcc --offload-arch=gfx950); "
                                     "there is no CPU fallback")
    L = C.CDLL(_LIB_PATH)
    L.tsdf_abi_version.restype = C.c_int
    if L.tsdf_abi_version() != ABI_VERSION:      # struct layouts differ between versions: refuse a mismatched library
        raise TsdfError(E_BADARG, f"{_LIB_PATH} has ABI version {L.tsdf_abi_version()}, this binding is for {ABI_VERSION}: rebuild (`make`)")
    H = C.c_void_p
    dp, fp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    sig = {
        "tsdf_abi_version": (C.c_int, []),
        "tsdf_default_config": (None, [C.POINTER(Config)]),
        "tsdf_create": (C.c_int, [C.POINTER(Config), C.POINTER(H)]),
        "tsdf_destroy": (None, [H]),
        "tsdf_last_error": (C.c_char_p, [H]),
        "tsdf_strerror": (C.c_char_p, [C.c_int]),
        "tsdf_get_config": (C.c_int, [H, C.POINTER(Config)]),
        "tsdf_set_intrinsics": (C.c_int, [H, dp]),
        "tsdf_set_camera_transformation": (C.c_int, [H, dp, dp]),
        "tsdf_set_tracker_params": (C.c_int, [H, C.c_int32, C.c_float, C.c_float, C.c_float]),
        "tsdf_frame_serial": (C.c_int64, [H]),
        "tsdf_device_frame_released": (C.c_int64, [H]),
        "tsdf_queue_frame": (C.c_int, [H, fp, fp, u8p, C.c_int32, C.c_int32]),
        "tsdf_queue_frame_aos": (C.c_int, [H, C.c_void_p, C.c_void_p, C.POINTER(AosLayout), C.c_int32, C.c_int32]),
        "tsdf_next_frame": (C.c_int, [H]),
        "tsdf_get_pose": (C.c_int, [H, dp, dp, dp, dp]),
        "tsdf_set_frame": (C.c_int, [H, fp, fp, u8p, C.c_int32, C.c_int32]),
        "tsdf_set_frame_device": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
        "tsdf_queue_frame_device": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
        "tsdf_set_frame_aos": (C.c_int, [H, C.c_void_p, C.c_void_p, C.POINTER(AosLayout), C.c_int32, C.c_int32]),
        "tsdf_track_aos": (C.c_int, [H, C.c_void_p, C.POINTER(AosLayout), C.c_int32, C.c_int32, C.POINTER(TrackStats)]),
        "tsdf_track_frame_aos": (C.c_int, [H, C.c_void_p, C.c_void_p, C.POINTER(AosLayout), C.c_int32, C.c_int32, C.POINTER(TrackStats)]),
        "tsdf_integrate_aos": (C.c_int, [H, C.c_void_p, C.c_void_p, C.POINTER(AosLayout), C.c_int32, C.c_int32, C.POINTER(IntegrateStats)]),
        "tsdf_default_preproc": (None, [C.POINTER(PreprocParams)]),
        "tsdf_set_depth_frame": (C.c_int, [H, C.POINTER(C.c_uint16), fp, u8p, C.c_int32, C.c_int32, C.POINTER(PreprocParams)]),
        "tsdf_queue_depth_frame": (C.c_int, [H, C.POINTER(C.c_uint16), fp, u8p, C.c_int32, C.c_int32, C.POINTER(PreprocParams)]),
        "tsdf_get_preprocessed": (C.c_int, [H, fp, fp]),
        "tsdf_integrate": (C.c_int, [H, C.POINTER(IntegrateStats)]),
        "tsdf_track": (C.c_int, [H, C.POINTER(TrackStats)]),
        "tsdf_accumulate": (C.c_int, [H, dp, dp, C.POINTER(AccumStats)]),
        "tsdf_gn_update": (C.c_int, [H, dp, dp, dp, ip]),
        "tsdf_track_and_integrate": (C.c_int, [H, C.c_int32, C.c_void_p, C.c_void_p]),
        "tsdf_sample": (C.c_int, [H, dp, C.c_int32, fp, ip]),
        "tsdf_mesh_extract": (C.c_int, [H, C.c_float, C.c_int32, C.POINTER(C.c_int64)]),
        "tsdf_mesh_read": (C.c_int, [H, fp, fp, C.c_int64]),
        "tsdf_mesh_device": (C.c_int, [H, C.POINTER(fp), C.POINTER(fp), C.POINTER(C.c_int64)]),
        "tsdf_download": (C.c_int, [H, fp, fp]),
        "tsdf_upload": (C.c_int, [H, fp, fp]),
        "tsdf_download_color": (C.c_int, [H, fp, fp, fp, fp]),
        "tsdf_upload_color": (C.c_int, [H, fp, fp, fp, fp]),
        "tsdf_upload_with_halo": (C.c_int, [H, fp, fp]),
        "tsdf_upload_color_with_halo": (C.c_int, [H, fp, fp, fp, fp]),
        "tsdf_reset": (C.c_int, [H]),
        "tsdf_save": (C.c_int, [H, C.c_char_p]),
        "tsdf_load": (C.c_int, [H, C.c_char_p]),
        "tsdf_slab_range": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, ip, ip]),
        "tsdf_cyclic_range": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, ip, ip, ip]),
        "tsdf_slab_range_weighted": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, ip, ip]),
        "tsdf_frustum_layer_weights": (C.c_int, [C.POINTER(Config), dp, C.c_int32, C.c_int32, dp, dp, C.c_float, dp]),
        "tsdf_halo_for": (C.c_int32, [C.POINTER(Config), C.c_float]),
        "tsdf_comm_unique_id": (C.c_int, [C.c_void_p]),
        "tsdf_comm_init": (C.c_int, [H, C.c_int32, C.c_int32, C.c_void_p]),
        "tsdf_comm_init_shm": (C.c_int, [H, C.c_int32, C.c_int32, C.c_char_p]),
        "tsdf_comm_init_peer": (C.c_int, [H, C.c_int32, C.c_int32, C.c_char_p]),
        "tsdf_comm_finalize": (C.c_int, [H]),
        "tsdf_set_allreduce_hook": (C.c_int, [H, ALLREDUCE_FN, C.c_void_p]),
        "tsdf_allreduce": (C.c_int, [H, dp, C.c_int32]),
        "tsdf_host_set_pose": (C.c_int, [dp, dp, dp, dp]),
        "tsdf_host_perturbed_rotations": (C.c_int, [dp, C.c_float, dp]),
        "tsdf_host_gn_step": (C.c_int, [dp, dp, dp, dp, C.c_float, dp, ip]),
        "tsdf_set_timing": (C.c_int, [H, C.c_int32]),
        "tsdf_read_timing": (C.c_int, [H, C.POINTER(Timing), C.c_int32]),
        "tsdf_read_counters": (C.c_int, [H, C.POINTER(Counters), C.c_int32]),
        "tsdf_synchronize": (C.c_int, [H]),
        "tsdf_stream": (C.c_void_p, [H]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)          # AttributeError here = ABI drift: fail loudly
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def default_config(**overrides) -> Config:
    cfg = Config()
    lib().tsdf_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if k == "origin":
            cfg.origin = (C.c_double * 3)(*[float(x) for x in v])
        else:
            if not hasattr(cfg, k):
                raise AttributeError(f"tsdf_config has no field {k}")
            setattr(cfg, k, v)
    return cfg


def slab_range(m: int, nranks: int, rank: int):
    x0, x1 = C.c_int32(), C.c_int32()
    rc = lib().tsdf_slab_range(m, nranks, rank, C.byref(x0), C.byref(x1))
    if rc:
        raise TsdfError(rc, "tsdf_slab_range: bad argument")
    return int(x0.value), int(x1.value)


def cyclic_range(m: int, nranks: int, rank: int, halo: int, block: int = 0):
    """tsdf_cyclic_range: (x0, x1, stride) of the block-cyclic placement -- SDF(m, slab=(x0, x1), halo=halo, slab_stride=stride).
    block = 0: two blocks per rank (m / (2 nranks), widened until the halo fits).  Raises when nothing fits."""
    x0, x1, st = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib().tsdf_cyclic_range(m, nranks, rank, halo, block, C.byref(x0), C.byref(x1), C.byref(st))
    if rc:
        raise TsdfError(rc, f"tsdf_cyclic_range: no block-cyclic placement for m={m}, {nranks} ranks, halo {halo}, block {block}")
    return int(x0.value), int(x1.value), int(st.value)


def slab_range_weighted(m: int, nranks: int, rank: int, halo: int, layer_weight):
    """tsdf_slab_range_weighted: slabs of equal expected work (weights per x layer, e.g. from frustum_layer_weights)."""
    w = np.ascontiguousarray(layer_weight, dtype=np.float64)
    if w.size != m:
        raise ValueError(f"expected {m} layer weights, got {w.size}")
    x0, x1 = C.c_int32(), C.c_int32()
    rc = lib().tsdf_slab_range_weighted(m, nranks, rank, int(halo), _dptr(w), C.byref(x0), C.byref(x1))
    if rc:
        raise TsdfError(rc, "tsdf_slab_range_weighted: bad argument")
    return int(x0.value), int(x1.value)


def frustum_layer_weights(cfg: Config, K, width, height, rot, trans, max_depth=5.0, weights=None):
    """tsdf_frustum_layer_weights: expected integration work per x layer for one pose, added to `weights`."""
    w = np.zeros(int(cfg.m)) if weights is None else weights
    k, r, t = _d(K, 9), _d(rot, 9), _d(trans, 3)
    rc = lib().tsdf_frustum_layer_weights(C.byref(cfg), _dptr(k), int(width), int(height), _dptr(r), _dptr(t), float(max_depth), _dptr(w))
    if rc:
        raise TsdfError(rc, "tsdf_frustum_layer_weights: bad argument")
    return w


def path_layer_weights(cfg: Config, K, width, height, rots, transs, max_depth=5.0, max_poses=64):
    """Expected integration work per x layer over a PLANNED camera path: tsdf_frustum_layer_weights accumulated over (at
    most max_poses, evenly spaced) poses of the path.  Slabs cut on these weights (slab_range_weighted) stay balanced along
    the whole path, not only while the camera keeps its first view."""
    n = len(rots)
    if n == 0:
        raise ValueError("path_layer_weights: empty path")
    picks = sorted({int(round(i * (n - 1) / max(1, min(n, max_poses) - 1))) for i in range(min(n, max_poses))})
    w = np.zeros(int(cfg.m))
    for i in picks:
        frustum_layer_weights(cfg, K, width, height, rots[i], transs[i], max_depth, w)
    return w


def slab_cuts_for_path(cfg: Config, K, width, height, rots, transs, nranks, halo, max_depth=5.0, max_poses=64):
    """Static x-slab boundaries [c_0 = 0, c_1, ..., c_nranks = m] for a PLANNED camera path.

    Every rank integrates (and every Gauss-Newton pass waits for) the busiest rank of the frame, so what a cut costs over a
    path is the path-average of max_r work(r, pose) -- not the work summed over the path, which a camera that sweeps across
    the volume balances however badly the single frames are split.  Start: the cut of equal accumulated work
    (slab_range_weighted on path_layer_weights); then a local search moves one boundary at a time while that average goes
    down.  work(r, pose) = frustum weights of the layers the rank STORES (slab + halo).  Deterministic: every rank of a job
    computes the same boundaries.  (over the whole fr1/plant path equal-thickness slabs are within 3 %
    of this optimum at 8 ranks, and the cut for the FIRST pose alone is 57 % worse.)"""
    m, n = int(cfg.m), int(nranks)
    if n <= 1:
        return [0, m]
    npose = len(rots)
    picks = sorted({int(round(i * (npose - 1) / max(1, min(npose, max_poses) - 1))) for i in range(min(npose, max_poses))})
    per_pose = np.stack([frustum_layer_weights(cfg, K, width, height, rots[i], transs[i], max_depth) for i in picks])
    pre = np.concatenate([np.zeros((len(picks), 1)), np.cumsum(per_pose, axis=1)], axis=1)

    def objective(c):
        cost = np.stack([pre[:, min(m, c[r + 1] + halo)] - pre[:, max(0, c[r] - halo)] for r in range(n)], axis=1)
        return float(cost.max(axis=1).mean())
    total = per_pose.sum(axis=0)
    best = [slab_range_weighted(m, n, r, halo, total)[0] for r in range(n)] + [m]
    uni = [slab_range(m, n, r)[0] for r in range(n)] + [m]
    if objective(uni) < objective(best):
        best = uni
    fb = objective(best)
    improved = True
    while improved:
        improved = False
        for i in range(1, n):
            for d in (-16, -4, -1, 1, 4, 16):
                c = list(best)
                c[i] += d
                if c[i] <= c[i - 1] or c[i] >= c[i + 1]:
                    continue
                f = objective(c)
                if f < fb * (1.0 - 1e-12):
                    best, fb, improved = c, f, True
    return [int(x) for x in best]


def halo_for(cfg: Config, max_range: float) -> int:
    return int(lib().tsdf_halo_for(C.byref(cfg), float(max_range)))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def host_set_pose(rot, trans):
    """(rot_inv, rot_inv_trans) as CameraTracking::set_camera_transformation computes them (host only)."""
    r, t, ri, rit = _d(rot, 9), _d(trans, 3), np.zeros(9), np.zeros(3)
    rc = lib().tsdf_host_set_pose(_dptr(r), _dptr(t), _dptr(ri), _dptr(rit))
    if rc:
        raise TsdfError(rc, "tsdf_host_set_pose")
    return ri.reshape(3, 3), rit


def host_perturbed_rotations(rot, w_h):
    r, out = _d(rot, 9), np.zeros(54)
    rc = lib().tsdf_host_perturbed_rotations(_dptr(r), float(w_h), _dptr(out))
    if rc:
        raise TsdfError(rc, "tsdf_host_perturbed_rotations")
    return out.reshape(6, 3, 3)


def host_gn_step(rot, trans, A, b, max_twist_diff=0.001):
    """One Gauss-Newton pose update on the host: returns (rot, trans, twist, stop); raises on a singular system."""
    r, t = _d(rot, 9).copy(), _d(trans, 3).copy()
    a, bb, tw, stop = _d(A, 36), _d(b, 6), np.zeros(6), C.c_int32(0)
    rc = lib().tsdf_host_gn_step(_dptr(r), _dptr(t), _dptr(a), _dptr(bb), float(max_twist_diff), _dptr(tw), C.byref(stop))
    if rc:
        raise TsdfError(rc, lib().tsdf_strerror(rc).decode())
    return r.reshape(3, 3), t, tw, bool(stop.value)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a, n):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


class SDF:
    """The reference's ``class SDF`` (sdf.h:35-186) with the volume resident in HBM.

    ``SDF(m, width, height, depth, sdf_origin, distance_delta, distance_epsilon)`` as in sdf.h:78-79;
    keyword extras select colour lanes, the x-slab this rank owns and the device.
    """

    def __init__(self, m=256, width=6.0, height=6.0, depth=3.5, sdf_origin=(-3.0, -3.0, -0.5),
                 distance_delta=0.3, distance_epsilon=0.025, *, with_color=True, slab=None, halo=0, slab_stride=0,
                 device=0, stale_carry=True, carry_threads=1, gn_max_iter=20, max_twist_diff=0.001, v_h=1.0, w_h=0.01,
                 pixel_stride=3):
        L = lib()
        cfg = default_config(m=int(m), width=float(width), height=float(height), depth=float(depth),
                             origin=sdf_origin, delta=float(distance_delta), epsilon=float(distance_epsilon),
                             with_color=1 if with_color else 0, halo=int(halo), device=int(device),
                             stale_carry=1 if stale_carry else 0, carry_threads=int(carry_threads), gn_max_iter=int(gn_max_iter),
                             max_twist_diff=float(max_twist_diff), v_h=float(v_h), w_h=float(w_h),
                             pixel_stride=int(pixel_stride))
        if slab is not None:
            cfg.slab_x0, cfg.slab_x1 = int(slab[0]), int(slab[1])
        cfg.slab_stride = int(slab_stride)      # > 0: block-cyclic placement, blocks [x0 + b*stride, x1 + b*stride) (tsdf.h)
        self._h = C.c_void_p()
        rc = L.tsdf_create(C.byref(cfg), C.byref(self._h))
        if rc:
            self._h = None
            raise TsdfError(rc, L.tsdf_last_error(None).decode())
        got = Config()
        L.tsdf_get_config(self._h, C.byref(got))
        self.cfg = got
        self.m = int(got.m)
        self.m_div_width = float(np.float32(got.m) / np.float32(got.width))
        self.m_div_height = float(np.float32(got.m) / np.float32(got.height))
        self.m_div_depth = float(np.float32(got.m) / np.float32(got.depth))
        self._keep = []          # callbacks / borrowed tensors kept alive

    # -- plumbing
    def _check(self, rc):
        if rc:
            raise TsdfError(rc, lib().tsdf_last_error(self._h).decode() or lib().tsdf_strerror(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().tsdf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def slab(self):
        return int(self.cfg.slab_x0), int(self.cfg.slab_x1)

    def get_number_of_voxels(self):
        return self.m ** 3

    # -- frames (the cloud_filtered / normals arguments of the reference's hot calls)
    def set_frame(self, xyz, normals=None, rgb=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if xyz.ndim != 3 or xyz.shape[2] != 3:
            raise ValueError("xyz must be (height, width, 3)")
        h, w = xyz.shape[:2]
        nptr = cptr = None
        if normals is not None:
            normals = np.ascontiguousarray(normals, dtype=np.float32)
            assert normals.shape == xyz.shape
            nptr = _fptr(normals)
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            assert rgb.shape == xyz.shape
            cptr = rgb.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(lib().tsdf_set_frame(self._h, _fptr(xyz), nptr, cptr, w, h))

    def set_depth_frame(self, depth, rgb=None, **params):
        """Raw depth image (uint16 with depth_scale, or float32 metres) -> GPU back-projection + bilateral filter +
        normals -> current frame.  Keyword overrides: depth_scale, sigma_s, sigma_r, radius, normal_radius,
        max_depth_change.  Needs the intrinsics (CameraTracking.set_K) first."""
        self._depth_frame(lib().tsdf_set_depth_frame, depth, rgb, params)

    def queue_depth_frame(self, depth, rgb=None, **params):
        """tsdf_queue_depth_frame: the same through the two-deep frame queue (see queue_frame); the arrays are borrowed
        until next_frame returns (references are kept that long)."""
        keep = self._depth_frame(lib().tsdf_queue_depth_frame, depth, rgb, params)
        self._queued_keep = (getattr(self, "_queued_keep", None) or []) + [keep]

    def _depth_frame(self, entry, depth, rgb, params):
        pp = PreprocParams()
        lib().tsdf_default_preproc(C.byref(pp))
        for k, v in params.items():
            if not hasattr(pp, k):
                raise AttributeError(f"tsdf_preproc_params has no field {k}")
            setattr(pp, k, v)
        depth = np.ascontiguousarray(depth)
        h, w = depth.shape
        d16 = df = None
        if depth.dtype == np.uint16:
            d16 = depth.ctypes.data_as(C.POINTER(C.c_uint16))
        else:
            depth = np.ascontiguousarray(depth, dtype=np.float32)
            df = _fptr(depth)
        cptr = None
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            assert rgb.shape == (h, w, 3)
            cptr = rgb.ctypes.data_as(C.POINTER(C.c_uint8))
        self._check(entry(self._h, d16, df, cptr, w, h, C.byref(pp)))
        self._frame_shape = (h, w)
        return (depth, rgb)

    def get_preprocessed(self):
        h, w = self._frame_shape
        xyz = np.empty((h, w, 3), dtype=np.float32)
        nrm = np.empty((h, w, 3), dtype=np.float32)
        self._check(lib().tsdf_get_preprocessed(self._h, _fptr(xyz), _fptr(nrm)))
        return xyz, nrm

    def set_frame_aos(self, points=None, normals=None):
        """The frame as arrays of point structs (numpy structured arrays of shape (height, width), e.g. of dtype
        PCL_POINT_XYZRGB / PCL_NORMAL): fields x, y, z (consecutive floats) and optionally r, g, b (bytes); normals with
        normal_x, normal_y, normal_z.  points=None keeps the xyz / rgb of the current host frame and adds the normals."""
        lay = AosLayout(0, 0, -1, -1, -1, 0, 0)
        shape = None
        pp = nn = None
        if points is not None:
            points = np.ascontiguousarray(points)
            f = points.dtype.fields
            if not (f["y"][1] == f["x"][1] + 4 and f["z"][1] == f["x"][1] + 8):
                raise ValueError("x, y, z must be consecutive floats")
            lay.point_stride, lay.xyz_offset = points.dtype.itemsize, f["x"][1]
            if all(k in f for k in "rgb"):
                lay.r_offset, lay.g_offset, lay.b_offset = f["r"][1], f["g"][1], f["b"][1]
            shape = points.shape
            pp = C.c_void_p(points.ctypes.data)
        if normals is not None:
            normals = np.ascontiguousarray(normals)
            f = normals.dtype.fields
            if not (f["normal_y"][1] == f["normal_x"][1] + 4 and f["normal_z"][1] == f["normal_x"][1] + 8):
                raise ValueError("normal_x, normal_y, normal_z must be consecutive floats")
            lay.normal_stride, lay.normal_offset = normals.dtype.itemsize, f["normal_x"][1]
            if shape is not None and normals.shape != shape:
                raise ValueError("points and normals differ in shape")
            shape = normals.shape
            nn = C.c_void_p(normals.ctypes.data)
        if shape is None or len(shape) != 2:
            raise ValueError("organized clouds of shape (height, width) are needed")
        self._check(lib().tsdf_set_frame_aos(self._h, pp, nn, C.byref(lay), shape[1], shape[0]))
        self._frame_shape = tuple(shape)

    @staticmethod
    def _aos_layout(points=None, normals=None):
        lay = AosLayout(0, 0, -1, -1, -1, 0, 0)
        if points is not None:
            f = points.dtype.fields
            lay.point_stride, lay.xyz_offset = points.dtype.itemsize, f["x"][1]
            if all(k in f for k in "rgb"):
                lay.r_offset, lay.g_offset, lay.b_offset = f["r"][1], f["g"][1], f["b"][1]
        if normals is not None:
            g = normals.dtype.fields
            lay.normal_stride, lay.normal_offset = normals.dtype.itemsize, g["normal_x"][1]
        return lay

    def track_aos(self, points, normals=None):
        """tsdf_track_aos = CameraTracking::estimate_new_position(sdf, cloud) on an array of point structs (C-contiguous
        structured array of shape (height, width)): samples first, the rest of the cloud staged under the passes.  With
        `normals` (tsdf_track_frame_aos) the whole frame is staged and packed under the passes: follow with update()."""
        if not points.flags["C_CONTIGUOUS"] or points.ndim != 2 or (normals is not None and not normals.flags["C_CONTIGUOUS"]):
            raise ValueError("an organized C-contiguous cloud of shape (height, width) is needed")
        lay = self._aos_layout(points, normals)
        st = TrackStats()
        if normals is None:
            rc = lib().tsdf_track_aos(self._h, C.c_void_p(points.ctypes.data), C.byref(lay), points.shape[1], points.shape[0], C.byref(st))
        else:
            rc = lib().tsdf_track_frame_aos(self._h, C.c_void_p(points.ctypes.data), C.c_void_p(normals.ctypes.data), C.byref(lay),
                                            points.shape[1], points.shape[0], C.byref(st))
        self._check(rc)
        return {"iterations": int(st.iterations), "stopped": int(st.stopped), "n_terms_last": int(st.n_terms_last),
                "last_twist": np.array(st.last_twist)}

    def update_aos(self, points, normals, want_stats=False):
        """tsdf_integrate_aos = SDF::update(tracker, cloud, normals); points may be None (= the cloud that was tracked)."""
        if not normals.flags["C_CONTIGUOUS"] or (points is not None and not points.flags["C_CONTIGUOUS"]):
            raise ValueError("C-contiguous clouds are needed")
        lay = self._aos_layout(points, normals)
        st = IntegrateStats()
        self._check(lib().tsdf_integrate_aos(self._h, C.c_void_p(points.ctypes.data) if points is not None else None,
                                             C.c_void_p(normals.ctypes.data), C.byref(lay), normals.shape[1], normals.shape[0],
                                             C.byref(st) if want_stats else None))
        return _struct_dict(st) if want_stats else None

    def queue_frame(self, xyz, normals=None, rgb=None):
        """tsdf_queue_frame: stage a coming frame while the current one is tracked / integrated (up to two frames may wait:
        the second gives a frame's staging and copy two frame times).  The arrays are borrowed until the next_frame()
        that makes this frame current returns (they are kept alive here); they must already be C-contiguous float32 / uint8."""
        for a, dt in ((xyz, np.float32), (normals, np.float32), (rgb, np.uint8)):
            if a is not None and not (a.flags["C_CONTIGUOUS"] and a.dtype == dt):
                raise ValueError("queue_frame borrows the buffers: C-contiguous float32 / uint8 arrays are needed")
        h, w = xyz.shape[:2]
        keep = (xyz, normals, rgb)
        self._check(lib().tsdf_queue_frame(self._h, _fptr(xyz), _fptr(normals) if normals is not None else None,
                                           rgb.ctypes.data_as(C.POINTER(C.c_uint8)) if rgb is not None else None, w, h))
        # only now: a refused call must not drop the references of the frames that ARE queued (up to two, oldest first)
        self._queued_keep = (getattr(self, "_queued_keep", None) or []) + [keep]

    def queue_frame_aos(self, points, normals=None):
        """tsdf_queue_frame_aos: as queue_frame, for arrays of point structs (see set_frame_aos)."""
        lay = AosLayout(0, 0, -1, -1, -1, 0, 0)
        f = points.dtype.fields
        lay.point_stride, lay.xyz_offset = points.dtype.itemsize, f["x"][1]
        if all(k in f for k in "rgb"):
            lay.r_offset, lay.g_offset, lay.b_offset = f["r"][1], f["g"][1], f["b"][1]
        nn = None
        if normals is not None:
            g = normals.dtype.fields
            lay.normal_stride, lay.normal_offset = normals.dtype.itemsize, g["normal_x"][1]
            nn = C.c_void_p(normals.ctypes.data)
        if not (points.flags["C_CONTIGUOUS"] and (normals is None or normals.flags["C_CONTIGUOUS"])):
            raise ValueError("queue_frame_aos borrows the buffers: C-contiguous arrays are needed")
        keep = (points, normals)
        self._check(lib().tsdf_queue_frame_aos(self._h, C.c_void_p(points.ctypes.data), nn, C.byref(lay), points.shape[1], points.shape[0]))
        self._queued_keep = (getattr(self, "_queued_keep", None) or []) + [keep]

    def next_frame(self):
        """tsdf_next_frame: the oldest queued frame becomes the current one; its buffers are the caller's again."""
        self._check(lib().tsdf_next_frame(self._h))
        self._queued_keep = (getattr(self, "_queued_keep", None) or [])[1:]

    def set_frame_device(self, d_xyz, d_normals, d_rgb, width, height, keep=None):
        """Borrow device pointers (ints, e.g. torch.Tensor.data_ptr()) of images already in HBM.  The frame is packed
        inside its integrate launch (asynchronous), so `keep` is held until the set_frame* call after the next one."""
        self._keep = getattr(self, "_keep", [])[-1:] + [keep]
        self._check(lib().tsdf_set_frame_device(self._h, C.c_void_p(d_xyz), C.c_void_p(d_normals or 0),
                                                C.c_void_p(d_rgb or 0), int(width), int(height)))

    def frame_serial(self):
        """tsdf_frame_serial: frames made current so far."""
        return int(lib().tsdf_frame_serial(self._h))

    def device_frame_released(self):
        """tsdf_device_frame_released: the library no longer reads the device planes of any frame with a serial up to
        this one (non-blocking)."""
        return int(lib().tsdf_device_frame_released(self._h))

    def queue_frame_device(self, d_xyz, d_normals, d_rgb, width, height, keep=None):
        """tsdf_queue_frame_device: queue a frame that is already in HBM (device pointers as ints); the current frame's
        integrate launch packs it.  The buffers must stay valid until the frame after this one has been made current
        (`keep` holds references that long)."""
        self._check(lib().tsdf_queue_frame_device(self._h, C.c_void_p(d_xyz), C.c_void_p(d_normals or 0),
                                                  C.c_void_p(d_rgb or 0), int(width), int(height)))
        self._queued_keep_dev = getattr(self, "_queued_keep_dev", [])[-1:] + [keep]

    # -- SDF::update(camera_tracking, cloud_filtered, normals), sdf.h:161-163
    def update(self, camera_tracking=None, cloud_filtered=None, normals=None, rgb=None, want_stats=True):
        if cloud_filtered is not None:
            self.set_frame(cloud_filtered, normals, rgb)
        if want_stats:
            st = IntegrateStats()
            self._check(lib().tsdf_integrate(self._h, C.byref(st)))
            return _struct_dict(st)
        self._check(lib().tsdf_integrate(self._h, None))
        return None

    # -- SDF::interpolate_distance, sdf.h:86 (batched; voxel coordinates like the reference's argument)
    def interpolate_distance(self, voxel_coordinates, raw=False):
        """tsdf_sample.  raw=True returns the int32 flags as they come: 1 interpolated, 0 not (the reference's
        is_interpolated = false), -1 a corner lies in the grid but outside the layers this handle stores."""
        v = np.ascontiguousarray(voxel_coordinates, dtype=np.float64).reshape(-1, 3)
        val = np.zeros(len(v), dtype=np.float32)
        ok = np.zeros(len(v), dtype=np.int32)
        self._check(lib().tsdf_sample(self._h, _dptr(v), len(v), _fptr(val), ok.ctypes.data_as(C.POINTER(C.c_int32))))
        return val, (ok if raw else ok.astype(bool))

    # -- the visualiser thread's mesh (sdf.cpp:317-391; pcl::MarchingCubesSDF::performReconstruction)
    def mesh(self, iso_level=0.0, with_color=False, read=True):
        """Marching cubes on the GPU.  Returns (n_tri, 3, 3) float32 vertices in the grid-local frame of the
        reference's cloud [, (n_tri, 3, 4) float32 colours]; read=False leaves the result in HBM and returns
        the triangle count."""
        n = C.c_int64(0)
        self._check(lib().tsdf_mesh_extract(self._h, iso_level, 1 if with_color else 0, C.byref(n)))
        if not read:
            return n.value
        v = np.empty((n.value, 3, 3), dtype=np.float32)
        c = np.empty((n.value, 3, 4), dtype=np.float32) if with_color else None
        self._check(lib().tsdf_mesh_read(self._h, _fptr(v), _fptr(c) if with_color else None, n.value))
        return (v, c) if with_color else v

    # -- host mirrors of D/W (what the reference hands to its mesher, sdf.cpp:47-49)
    def owned_layers(self):
        """x layers this handle owns: the slab's, or the sum over the blocks of a block-cyclic handle (owned_x() lists them)."""
        return len(self.owned_x())

    def owned_x(self):
        """global x indices of the owned layers, in the order download() returns them"""
        x0, x1, st = int(self.cfg.slab_x0), int(self.cfg.slab_x1), int(self.cfg.slab_stride)
        if st <= 0:
            return np.arange(x0, x1)
        return np.concatenate([np.arange(x0 + b, x1 + b) for b in range(0, self.m - x0, st)])

    def download(self):
        n = self.owned_layers() * self.m * self.m
        D = np.empty(n, dtype=np.float32)
        W = np.empty(n, dtype=np.float32)
        self._check(lib().tsdf_download(self._h, _fptr(D), _fptr(W)))
        return D, W

    def upload(self, D, W):
        D = np.ascontiguousarray(D, dtype=np.float32).reshape(-1)
        W = np.ascontiguousarray(W, dtype=np.float32).reshape(-1)
        n = self.owned_layers() * self.m * self.m
        assert D.size == n and W.size == n
        self._check(lib().tsdf_upload(self._h, _fptr(D), _fptr(W)))

    def upload_with_halo(self, D_full, W_full):
        """Upload this rank's stored layers (slab + halo) out of full-volume host arrays."""
        m = self.m
        xs = max(0, self.cfg.slab_x0 - self.cfg.halo)
        xe = min(m, self.cfg.slab_x1 + self.cfg.halo)
        D = np.ascontiguousarray(np.asarray(D_full, dtype=np.float32).reshape(m, m, m)[xs:xe]).reshape(-1)
        W = np.ascontiguousarray(np.asarray(W_full, dtype=np.float32).reshape(m, m, m)[xs:xe]).reshape(-1)
        self._check(lib().tsdf_upload_with_halo(self._h, _fptr(D), _fptr(W)))

    def upload_color_with_halo(self, Color_W_full, R_full, G_full, B_full):
        """Colour counterpart of upload_with_halo (full-volume host arrays in, this rank's stored layers up)."""
        m = self.m
        xs = max(0, self.cfg.slab_x0 - self.cfg.halo)
        xe = min(m, self.cfg.slab_x1 + self.cfg.halo)
        arrs = [np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(m, m, m)[xs:xe]).reshape(-1)
                for a in (Color_W_full, R_full, G_full, B_full)]
        self._check(lib().tsdf_upload_color_with_halo(self._h, *[_fptr(a) for a in arrs]))

    def download_color(self):
        n = self.owned_layers() * self.m * self.m
        out = [np.empty(n, dtype=np.float32) for _ in range(4)]
        self._check(lib().tsdf_download_color(self._h, *[_fptr(a) for a in out]))
        return tuple(out)

    def upload_color(self, Color_W, R, G, B):
        arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in (Color_W, R, G, B)]
        self._check(lib().tsdf_upload_color(self._h, *[_fptr(a) for a in arrs]))

    def reset(self):
        self._check(lib().tsdf_reset(self._h))

    def save(self, path):
        """Write this handle's stored layers (slab + halo): D, W (and colour) as a TSDFVOL2 checkpoint."""
        self._check(lib().tsdf_save(self._h, str(path).encode()))

    def load(self, path):
        """Restore the stored layers from a TSDFVOL2 file that covers them (this shard's own file, or a whole-volume one)."""
        self._check(lib().tsdf_load(self._h, str(path).encode()))

    # -- multi-GPU plumbing
    def comm_init(self, nranks, rank, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(lib().tsdf_comm_init(self._h, nranks, rank, buf))

    def comm_init_shm(self, nranks, rank, name: str):
        self._check(lib().tsdf_comm_init_shm(self._h, nranks, rank, name.encode()))

    def comm_init_peer(self, nranks, rank, name: str):
        self._check(lib().tsdf_comm_init_peer(self._h, nranks, rank, name.encode()))

    def comm_finalize(self):
        self._check(lib().tsdf_comm_finalize(self._h))

    def set_allreduce_hook(self, fn):
        """fn(np.ndarray[float64]) sums the array over ranks in place."""
        if fn is None:
            self._hook = None
            self._check(lib().tsdf_set_allreduce_hook(self._h, C.cast(None, ALLREDUCE_FN), None))
            return

        def _tramp(buf, n, _ctx):
            try:
                fn(np.ctypeslib.as_array(buf, shape=(n,)))
                return 0
            except Exception:
                return 1
        self._hook = ALLREDUCE_FN(_tramp)
        self._check(lib().tsdf_set_allreduce_hook(self._h, self._hook, None))

    def allreduce(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        self._check(lib().tsdf_allreduce(self._h, _dptr(a), a.size))
        return a

    # -- measurement
    def set_timing(self, on=True, track=None, period=1):
        """HIP-event timing: integrate/pack launches when ``on`` (every ``period``-th launch of a kind); tracker
        passes too when ``track`` (default: same as ``on``; per-pass events cost a few microseconds each)."""
        track = on if track is None else track
        self._check(lib().tsdf_set_timing(self._h, (1 if on else 0) | (2 if track else 0) | ((int(period) & 0xFF) << 8)))

    def read_timing(self, reset=False):
        t = Timing()
        self._check(lib().tsdf_read_timing(self._h, C.byref(t), 1 if reset else 0))
        return _struct_dict(t)

    def read_counters(self, reset=False):
        c = Counters()
        self._check(lib().tsdf_read_counters(self._h, C.byref(c), 1 if reset else 0))
        return _struct_dict(c)

    def synchronize(self):
        self._check(lib().tsdf_synchronize(self._h))

    @property
    def stream(self):
        return lib().tsdf_stream(self._h)


class CameraTracking:
    """The reference's ``class CameraTracking`` (camera_tracking.h:12-105).

    Constructor arguments follow the *definition* (camera_tracking.cpp:3-4):
    ``(gauss_newton_max_iteration, maximum_twist_diff, v_h, w_h, sdf)``; constants that differ from the ones the
    volume handle was created with reconfigure it (tsdf_set_tracker_params), as the reference's constructor accepts any.
    Pose state lives in the same native handle as the volume.
    """

    def __init__(self, gauss_newton_max_iteration=20, maximum_twist_diff=0.001, v_h=1.0, w_h=0.01, sdf: SDF = None):
        if sdf is None:
            raise ValueError("CameraTracking needs the SDF it tracks against")
        c = sdf.cfg
        want = (int(gauss_newton_max_iteration), np.float32(maximum_twist_diff), np.float32(v_h), np.float32(w_h))
        have = (int(c.gn_max_iter), np.float32(c.max_twist_diff), np.float32(c.v_h), np.float32(c.w_h))
        if want != have:
            sdf._check(lib().tsdf_set_tracker_params(sdf._h, want[0], C.c_float(want[1]), C.c_float(want[2]), C.c_float(want[3])))
            lib().tsdf_get_config(sdf._h, C.byref(sdf.cfg))
        self.sdf = sdf
        self.isKFilled = False

    def _pose(self):
        rot, trans, ri, rit = np.zeros(9), np.zeros(3), np.zeros(9), np.zeros(3)
        self.sdf._check(lib().tsdf_get_pose(self.sdf._h, _dptr(rot), _dptr(trans), _dptr(ri), _dptr(rit)))
        return rot.reshape(3, 3), trans, ri.reshape(3, 3), rit

    # public fields of camera_tracking.h:43-49
    @property
    def rot(self):
        return self._pose()[0]

    @property
    def trans(self):
        return self._pose()[1]

    @property
    def rot_inv(self):
        return self._pose()[2]

    @property
    def rot_inv_trans(self):
        return self._pose()[3]

    def set_K(self, K):
        """camera_info_cb (camera_tracking.cpp:22-36)."""
        k = _d(K, 9)
        self.K = k.reshape(3, 3).copy()
        self.sdf._check(lib().tsdf_set_intrinsics(self.sdf._h, _dptr(k)))
        self.isKFilled = True

    def set_camera_transformation(self, rot, trans):
        r, t = _d(rot, 9), _d(trans, 3)
        self.sdf._check(lib().tsdf_set_camera_transformation(self.sdf._h, _dptr(r), _dptr(t)))

    def estimate_new_position(self, sdf: SDF = None, point_cloud=None):
        """camera_tracking.h:101.  ``point_cloud`` = xyz (h, w, 3); None keeps the frame already set."""
        s = sdf or self.sdf
        if point_cloud is not None:
            s.set_frame(point_cloud)
        st = TrackStats()
        s._check(lib().tsdf_track(s._h, C.byref(st)))
        return _struct_dict(st)

    def accumulate(self):
        """One Gauss-Newton accumulation at the current pose: (A 6x6, b 6, stats), this rank's part only."""
        A, b, st = np.zeros(36), np.zeros(6), AccumStats()
        self.sdf._check(lib().tsdf_accumulate(self.sdf._h, _dptr(A), _dptr(b), C.byref(st)))
        return A.reshape(6, 6), b, _struct_dict(st)

    def gn_update(self, A, b):
        a, bb, tw, stop = _d(A, 36), _d(b, 6), np.zeros(6), C.c_int32(0)
        self.sdf._check(lib().tsdf_gn_update(self.sdf._h, _dptr(a), _dptr(bb), _dptr(tw), C.byref(stop)))
        return bool(stop.value), tw
