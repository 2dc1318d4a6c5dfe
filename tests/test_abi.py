"""CPU checks of the drop-in boundary: the C-ABI library loads here (no GPU), exports every symbol that
include/tsdf.h declares, and fails loudly -- never falls back -- when asked to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
import tracking_sdf_amd as ts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tsdf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsdf_[a-z0-9_]+)\s*\(", text)) - {"tsdf_allreduce_fn"})


def test_header_symbols_all_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 35
    L = ts.lib()
    for name in syms:
        assert hasattr(L, name), f"{name} declared in include/tsdf.h but not exported by libtsdf_hip.so"
    assert sorted(ts.ABI_SYMBOLS) == syms, "python binding and header disagree"
    assert L.tsdf_abi_version() == 4


def test_struct_layouts_match_header():
    # tsdf_config: 4+3*4 = 16, origin (8-aligned) 24, then 4*... -> check against a hand computation
    assert C.sizeof(ts.Config) == 104          # (slab_stride + a padding word in front of the 8-aligned end)
    assert ts.Config.origin.offset == 16 and ts.Config.delta.offset == 40 and ts.Config.carry_threads.offset == 72
    assert ts.Config.device.offset == 92
    assert C.sizeof(ts.IntegrateStats) == 24 and C.sizeof(ts.AccumStats) == 48
    assert C.sizeof(ts.TrackStats) == 64 and C.sizeof(ts.Timing) == 48 and C.sizeof(ts.Counters) == 80   # ABI 3: + track_passes_own_queue


def test_default_config_is_the_reference_constants():
    c = ts.default_config()
    assert (c.m, c.width, c.height, c.depth) == (256, 6.0, 6.0, 3.5)            # sdf_reconstruction.cpp:83-85
    assert list(c.origin) == [-3.0, -3.0, -0.5]
    assert c.delta == np.float32(0.3) and c.epsilon == np.float32(0.025)
    assert (c.gn_max_iter, c.v_h, c.pixel_stride) == (20, 1.0, 3)                # :88, camera_tracking.cpp:162
    assert c.max_twist_diff == np.float32(0.001) and c.w_h == np.float32(0.01)
    assert c.stale_carry == 1 and c.carry_threads == 1 and c.with_color == 1


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(ts.TsdfError) as ei:
        ts.SDF(32)
    assert ei.value.code in (ts.E_NO_DEVICE, ts.E_HIP)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_bad_arguments_are_errors_not_crashes():
    L = ts.lib()
    h = C.c_void_p()
    assert L.tsdf_create(None, C.byref(h)) == ts.E_BADARG
    cfg = ts.default_config(m=1)
    assert L.tsdf_create(C.byref(cfg), C.byref(h)) == ts.E_BADARG
    cfg = ts.default_config(m=64)
    cfg.slab_x0, cfg.slab_x1 = 40, 20
    assert L.tsdf_create(C.byref(cfg), C.byref(h)) == ts.E_BADARG
    assert L.tsdf_integrate(None, None) == ts.E_BADARG and L.tsdf_track(None, None) == ts.E_BADARG
    assert b"bad slab" in L.tsdf_last_error(None)
    assert L.tsdf_strerror(ts.E_HALO) == b"slab halo too small"


def test_slab_ranges_partition_the_axis():
    for m in (48, 64, 512, 2048):
        for n in (1, 2, 3, 4, 8):
            edges = [ts.slab_range(m, n, r) for r in range(n)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(n - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ts.TsdfError):
        ts.slab_range(64, 4, 4)


def test_cyclic_ranges_partition_the_axis_and_keep_the_blocks_apart():
    """tsdf_cyclic_range: the blocks [x0 + j stride, x1 + j stride) of all ranks cover every layer exactly once; a rank's
    stored ranges (block + halo) never overlap; block = 0 gives two blocks per rank where the halo allows; what cannot work
    is refused."""
    for m, n, halo in ((512, 8, 9), (512, 2, 9), (512, 3, 9), (128, 3, 5), (2048, 8, 24), (64, 2, 4), (256, 8, 6), (1024, 5, 14)):
        owner = np.full(m, -1)
        for r in range(n):
            x0, x1, st = ts.cyclic_range(m, n, r, halo)
            B = x1 - x0
            assert B & (B - 1) == 0 and m % B == 0 and st == n * B and x0 == r * B and st >= B + 2 * halo
            for a in range(x0, m, st):
                assert (owner[a:a + B] == -1).all()
                owner[a:a + B] = r
        assert (owner >= 0).all()
        if (n & (n - 1)) == 0 and (n - 1) * (m // (2 * n)) >= 2 * halo:
            assert B == m // (2 * n)                       # two blocks per rank
    assert ts.cyclic_range(512, 8, 3, 9, 16) == (48, 64, 128)              # a block size of the caller's choice
    for bad in ((96, 2, 0, 2, 0), (64, 2, 0, 40, 0), (64, 8, 0, 2, 16), (64, 1, 0, 2, 0), (64, 2, 2, 2, 0), (64, 2, 0, 2, 12)):
        with pytest.raises(ts.TsdfError):
            ts.cyclic_range(*bad)


def test_weighted_slab_ranges_partition_the_axis_and_balance_the_work():
    """tsdf_slab_range_weighted: contiguous non-empty slabs that partition the axis, deterministic; the largest cost
    (weights of the stored layers: slab + halo) is no larger than with equal slabs and within a layer's weight of the
    optimum a brute-force search finds; zero weights fall back to equal slabs.  tsdf_frustum_layer_weights: the initial
    pose looks along -y from the middle of the x extent, so the weights are symmetric in x and largest in the middle."""
    from tracking_sdf_amd import synth
    rng = np.random.default_rng(3)
    for m, n, halo in ((48, 4, 2), (64, 8, 3), (100, 3, 0), (33, 5, 4)):
        for trial in range(6):
            w = rng.random(m) ** 3 * (1 + 50 * (np.abs(np.arange(m) - rng.integers(m)) < m // 6))
            cuts = [ts.slab_range_weighted(m, n, r, halo, w) for r in range(n)]
            assert cuts[0][0] == 0 and cuts[-1][1] == m and all(b > a for a, b in cuts)
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(n - 1))
            assert cuts == [ts.slab_range_weighted(m, n, r, halo, w.copy()) for r in range(n)]
            pre = np.concatenate([[0.0], np.cumsum(w)])
            cost = lambda a, b: pre[min(m, b + halo)] - pre[max(0, a - halo)]
            got = max(cost(a, b) for a, b in cuts)
            uni = max(cost(*ts.slab_range(m, n, r)) for r in range(n))
            assert got <= uni * (1 + 1e-9)
            # brute force (dynamic programme over the cut positions)
            best = np.full((n + 1, m + 1), np.inf)
            best[0, 0] = 0.0
            for r in range(1, n + 1):
                for b in range(r, m - (n - r) + 1):
                    best[r, b] = min(max(best[r - 1, a], cost(a, b)) for a in range(r - 1, b))
            assert got <= best[n, m] * (1 + 1e-9) + 1e-12
    assert [ts.slab_range_weighted(64, 4, r, 2, np.zeros(64)) for r in range(4)] == [ts.slab_range(64, 4, r) for r in range(4)]
    with pytest.raises(ts.TsdfError):
        ts.slab_range_weighted(64, 4, 0, 2, -np.ones(64))
    with pytest.raises(ValueError):
        ts.slab_range_weighted(64, 4, 0, 2, np.ones(63))
    cfg = ts.default_config(m=128)
    K = synth.default_intrinsics(640, 480)
    W = ts.frustum_layer_weights(cfg, K, 640, 480, [[1, 0, 0], [0, 0, -1], [0, -1, 0]], [0, 0, 1], 5.0)
    assert W.shape == (128,) and np.all(W > 0) and np.allclose(W, W[::-1], rtol=1e-9) and W[64] > 5 * W[0]
    W2 = ts.frustum_layer_weights(cfg, K, 640, 480, [[1, 0, 0], [0, 0, -1], [0, -1, 0]], [0, 0, 1], 5.0, weights=W.copy())
    assert np.allclose(W2, 2 * W)                         # poses accumulate
    cuts = [ts.slab_range_weighted(128, 8, r, 4, W) for r in range(8)]
    sizes = [b - a for a, b in cuts]
    assert sizes[0] > 2 * sizes[3] and sizes == sizes[::-1]      # thick slabs at the edges, thin ones where the camera looks


def test_halo_covers_the_rotational_reach():
    cfg = ts.default_config(m=512)
    # w_h * range * m/width = 0.01 * 6 m * 85.33 voxel/m = 5.12 -> 6, + ceil(v_h) = 1, + 2 safety
    assert ts.halo_for(cfg, 6.0) == 9
    cfg = ts.default_config(m=2048)
    assert ts.halo_for(cfg, 5.0) == int(np.ceil(0.01 * 5.0 * 2048 / 6.0)) + 1 + 2


# ---- the product's host algebra against the oracle's (both restate Eigen; written independently)
def test_host_pose_algebra_matches_oracle():
    rng = np.random.default_rng(2)
    s = orc.SDF(8, 8.0, 8.0, 8.0, (0, 0, 0), 0.3, 0.025)
    t = orc.CameraTracking(s)
    for _ in range(20):
        q, _r = np.linalg.qr(rng.standard_normal((3, 3)))
        tr = rng.standard_normal(3)
        t.set_camera_transformation(q, tr)
        ri, rit = ts.host_set_pose(q, tr)
        assert np.array_equal(ri, t.rot_inv) and np.array_equal(rit, t.rot_inv_trans)
        assert np.array_equal(ts.host_perturbed_rotations(q, 0.01), t.perturbed_rotations())


def test_host_gn_step_matches_oracle_bit_for_bit():
    rng = np.random.default_rng(3)
    s = orc.SDF(8, 8.0, 8.0, 8.0, (0, 0, 0), 0.3, 0.025)
    t = orc.CameraTracking(s)
    for k in range(30):
        J = rng.standard_normal((200, 6))
        A = J.T @ J
        b = J.T @ (rng.standard_normal(200) * (1e-3 if k % 2 else 1e-1))
        rot0, trans0 = t.rot.copy(), t.trans.copy()
        stop_o, tw_o = t.gn_update(A, b)
        rot, trans, tw, stop = ts.host_gn_step(rot0, trans0, A, b, 0.001)
        assert stop == stop_o and np.array_equal(tw, tw_o)
        assert np.array_equal(rot, t.rot) and np.array_equal(trans, t.trans)
    with pytest.raises(ts.TsdfError) as ei:
        ts.host_gn_step(np.eye(3), np.zeros(3), np.zeros((6, 6)), np.ones(6))
    assert ei.value.code == ts.E_SINGULAR
    # signed stop rule (camera_tracking.cpp:216-224)
    assert ts.host_gn_step(np.eye(3), np.zeros(3), np.eye(6), -np.ones(6))[3] is True
    assert ts.host_gn_step(np.eye(3), np.zeros(3), np.eye(6), np.array([0, 0, 0, 0, 0, 0.002]))[3] is False


def test_missing_rccl_is_an_error_code_not_a_crash():
    """A box where librccl cannot be loaded: tsdf_comm_unique_id must return TSDF_E_COMM with the loader's
    message (the first version called dlerror() twice and built a std::string from NULL)."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import tracking_sdf_amd as ts\n"
        "L = ts.lib(); buf = C.create_string_buffer(128)\n"
        "rc = L.tsdf_comm_unique_id(buf)\n"
        "rc2 = L.tsdf_comm_unique_id(buf)\n"
        "print(rc, rc2, L.tsdf_last_error(None).decode())\n"
    )
    env = dict(os.environ, TSDF_RCCL_LIBRARY="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rc, rc2, msg = r.stdout.strip().split(" ", 2)
    assert int(rc) == ts.E_COMM and int(rc2) == ts.E_COMM
    assert "cannot load librccl" in msg and "nonexistent" in msg


def test_graft_entry_build_passes():
    """The driver's "does it build" check: __graft_entry__.build() (make all + ABI version + every declared symbol)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").build()


def test_library_and_stand_alone_tracker_code_object_carry_the_same_build_id():
    """lib/tsdf_track.hsaco (the tracker kernel for the optional AQL queue) is a target of its own and carries
    the hash of the tracker's sources + flags that the library carries; AqlQueue::init refuses any other code object."""
    import re
    libdir = os.path.join(ROOT, "tracking_sdf_amd", "lib")
    so = open(os.path.join(libdir, "libtsdf_hip.so"), "rb").read()
    co = open(os.path.join(libdir, "tsdf_track.hsaco"), "rb").read()
    ids = set(re.findall(rb"(?<![0-9a-f])[0-9a-f]{32}\x00", co))
    assert len(ids) == 1, ids
    the_id = next(iter(ids))[:-1]
    assert so.count(the_id) >= 2          # host side (track_kernel_build_id) and the library's own device code
    mk = open(os.path.join(ROOT, "Makefile")).read()
    assert ".DELETE_ON_ERROR" in mk and "$(HSACO): $(TRACK_DEPS)" in mk

