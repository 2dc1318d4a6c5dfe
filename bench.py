#!/usr/bin/env python3
"""bench.py -- frames/sec + ATE-RMSE of the tracking_sdf hot path on MI355X.

A "step" is one frame of the reference's per-frame hot path (sdf_reconstruction.cpp:69-74):
CameraTracking::estimate_new_position (<= 20 Gauss-Newton passes) followed by SDF::update, on a synthetic
depth stream rendered along the real fr1/plant ground-truth camera path (no TUM image data exists on the box).

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

  --gpus N > 1 without WORLD_SIZE in the environment: this process starts N rank processes itself (before
  anything touches a GPU), relays rank 0's JSON line and exits non-zero if any rank fails.  Under
  torch.distributed.run (WORLD_SIZE set) it is one rank.  One rank per GPU; the volume is sharded into N x-slabs +
  halo, one 30-double all-reduce of the normal equations per Gauss-Newton pass (in-library RCCL by default).

  --config  3 (default) the metric's workload: 512^3, 640x480, fixed volume -> strong scaling over N
            2           256^3, 640x480
            4           weak-scaling shape of "1024^3 on 8 GPUs": m = 1024 (N/8)^(1/3), 640x480
            5           weak-scaling shape of "2048^3 on 8 GPUs": m = 2048 (N/8)^(1/3), 1280x960
            (the x extent of a rank's slab shrinks as m grows, its voxel count stays m^3/N = const)

`value` times the hot path with every frame already resident in HBM.  The N = 1 default run adds, outside that
timed region: the PCIe-inclusive rates (`value_h2d_inclusive`: xyz + normals + rgb handed over as host buffers every
frame, SURVEY 8d's end-to-end definition; `value_depth_input_inclusive`: raw depth + rgb, pre-processed on the GPU),
the ATE over the whole 1246-frame sequence at 256^3 and at the benchmark m, the HBM traffic of the integrate launch
from two rocprofv3 --pmc child passes, a config-5-shaped leg, and the CPU baseline.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {     # config -> (m at the reference GPU count, reference GPU count, width, height, label)
    2: (256, 1, 640, 480, "config 2: fr1/plant path, 256^3, 640x480"),
    3: (512, 1, 640, 480, "config 3 / metric (BASELINE config 3 names fr1/desk: no fr1/desk trajectory exists in the reference tree or "
                          "on the box, so the metric's own fr1/plant path is run at its size): fr1/plant path, 512^3, 640x480"),
    4: (1024, 8, 640, 480, "config 4 shape: 1024^3 on 8 GPUs, 640x480, fr3 intrinsics (weak: m = 1024 (N/8)^(1/3))"),
    5: (2048, 8, 1280, 960, "config 5 shape: 2048^3 on 8 GPUs, 1280x960 (weak: m = 2048 (N/8)^(1/3))"),
}
# config 4 is a freiburg3 sequence -- its calibrated intrinsics (fx, fy, cx, cy) instead of the ROS default
FR3_K = ((535.4, 0.0, 320.1), (0.0, 539.2, 247.6), (0.0, 0.0, 1.0))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=sorted(WORKLOADS))
    ap.add_argument("--voxels", dest="m", type=int, default=None, help="voxels per axis (overrides the config's m)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--no-color", action="store_true", help="drop the colour lanes (sdf.cpp:294-304)")
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--frame-step", type=int, default=1, help="use every n-th 30 Hz pose")
    ap.add_argument("--max-range", type=float, default=6.0, help="metres; sizes the slab halo")
    ap.add_argument("--cyclic-block", type=int, default=0,
                    help="--slabs cyclic: x layers per block (0 = m / (2 N) rounded down to a power of two: two blocks per rank, "
                         "the best trade of halo work against balance in DESIGN 6.1's table)")
    ap.add_argument("--slabs", default="auto", choices=["auto", "cyclic", "path", "balanced", "uniform"],
                    help="N > 1: cyclic = BLOCK-CYCLIC placement (tsdf_config::slab_stride): rank r owns the blocks [r B + j N B, (r+1) B + j N B), "
                         "each stored with its halo -- every rank holds a share of every view, whatever the camera does (a plain slab's busiest "
                         "rank carries 0.32-0.34 of a frame's work at N = 8 along fr1/plant, a block-cyclic rank 0.22-0.25); "
                         "path = x-slabs of equal expected WORK over the PLANNED camera path (tsdf_slab_range_weighted on view-frustum "
                         "weights accumulated over the poses this run will visit: thin slabs where the camera looks, along the whole path); "
                         "auto (default) = of cyclic (where the volume allows it: m a power of two, blocks wider than the halo) and path, the "
                         "one whose busiest rank carries the smaller share of a frame's work over the planned path (it is known here: the "
                         "ground-truth trajectory), else uniform; "
                         "balanced = the weights of the reference's initial pose only (camera_tracking.cpp:5-7; round 5's default: good "
                         "for the first ~100 frames, worse than uniform 480 frames down the path); uniform = equal thickness (tsdf_slab_range)")
    ap.add_argument("--cpu-baseline-frames", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="main timed region only: no PCIe-inclusive legs, full-sequence ATE, PMC passes, weak leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes")
    ap.add_argument("--no-full-sequence", action="store_true")
    ap.add_argument("--no-weak-leg", action="store_true")
    ap.add_argument("--allreduce", choices=["rccl", "auto", "shm", "peer", "torch"], default="auto",
                    help="exchange step of a Gauss-Newton pass: auto (default) = every in-library step that passes its self-test runs "
                         "a short trial of the frame loop on this machine and the fastest is kept (a pass costs ~25 us: the exchange "
                         "step must be chosen by what it costs INSIDE a pass, on the node at hand -- DESIGN.md section 6); rccl = "
                         "in-library RCCL all-reduce (falls back to the shared-memory fan-in only if RCCL fails its self-test); shm = "
                         "host-side fan-in through a POSIX shared segment; peer = device-side exchange through HIP-IPC-mapped buffers "
                         "(tsdf_comm_init_peer)")
    ap.add_argument("--queue-ahead", type=int, default=2, choices=[1, 2],
                    help="frames waiting in the library's queue behind the current one in the host-frame legs (the queue takes "
                         "two; 1 = round 5's two-deep use)")
    ap.add_argument("--no-frame-queue", dest="frame_queue", action="store_false",
                    help="set every HBM-resident frame in front of its own tracker passes (tsdf_set_frame_device) instead of queueing frame "
                         "k+1 (tsdf_queue_frame_device) while frame k is processed.  Default since round 6: the queue -- frame k+1 is then "
                         "packed, sample list included, inside frame k's integrate launch (+3 %% frames/s, same bits)")
    ap.set_defaults(frame_queue=True)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the launcher plumbing (gloo lets several ranks share one GPU for testing)")
    ap.add_argument("--timing-period", type=int, default=4, help="HIP events around every n-th integrate/pack launch")
    ap.add_argument("--trajectory-out", default=None, help="write the estimated trajectory (TUM format)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rccl-under-gloo", action="store_true",
                    help="keep the in-library RCCL exchange step as a candidate under --dist-backend gloo (ranks sharing one GPU: "
                         "only with TSDF_RCCL_LIBRARY pointing at a library that accepts that, e.g. the tests' shared-memory mock)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher: N rank processes, started before anything here touches a GPU (torch is not even imported)

def launch_ranks(args):
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    rc = 0
    out0 = ""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if r == 0:
                    out0 = procs[0].stdout.read()
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            if pending:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = [ln for ln in out0.splitlines() if ln.startswith('{"metric"')]
    for ln in out0.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if rc == 0 and not lines:
        print("[bench] rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


# ----------------------------------------------------------------------------------------------------------------

def horn_rmse(est, gt):
    """ATE-RMSE after the rigid (proper rotation + translation) least-squares alignment of est onto gt.  Both
    trajectories live in the reference's mirrored world (synth.load_trajectory re-bases the ground truth onto the
    reference's det = -1 initial pose), so no reflection is needed here; tools/evaluate_ate.py has the
    reflection-tolerant form for files in the TUM frame."""
    if len(est) < 3:
        return float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1))))
    ce, cg = est.mean(0), gt.mean(0)
    H = (est - ce).T @ (gt - cg)
    U, _, Vt = np.linalg.svd(H)
    S = np.eye(3)
    if np.linalg.det(Vt.T @ U.T) < 0:
        S[2, 2] = -1
    R = Vt.T @ S @ U.T
    al = (est - ce) @ R.T + cg
    return float(np.sqrt(np.mean(np.sum((al - gt) ** 2, axis=1))))


def usable_cores():
    """Host threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def free_run_compare(args, ts, orc, K, frames, m, width, height, n, track_threads, update_threads, dev_index, with_global_coords,
                     time_limit_s=20.0):
    """The oracle (the reference's CPU path restated) and the HIP path on the SAME frames, both free-running from the
    reference's initial pose (every pose feeds the next integration), compared frame by frame and, at the end, voxel by
    voxel.  track_threads = the OpenMP thread count of the oracle's tracker = carry_threads of the HIP handle (the
    reference's carry state is thread-local, camera_tracking.cpp:148-159; 1 = SURVEY 8a8' canonical order).  Returns
    (timing of the oracle run, parity record).  The oracle is the checker here, after the timed region; never the product."""
    color = not args.no_color
    oo = orc.SDF(m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025, with_global_coords=with_global_coords)
    oo.track_exp_band()                      # which voxels took the exp() weight: the only ones allowed to differ
    ot = orc.CameraTracking(oo)
    ot.set_K(K)
    xyz, nrm, rgb = frames[0]
    n_upd = [oo.update(ot, orc.Cloud(xyz, nrm, rgb), with_color=color, threads=update_threads)]
    iters, stops, poses, rots = [], [], [ot.trans.copy()], [ot.rot.copy()]
    t_track = t_upd = 0.0
    done = 0
    t_all = time.perf_counter()
    for k in range(1, n + 1):
        xyz, nrm, rgb = frames[k]
        cloud = orc.Cloud(xyz, nrm, rgb)
        t0 = time.perf_counter()
        so = ot.estimate_new_position(oo, cloud, threads=track_threads, stale_carry=True)
        t1 = time.perf_counter()
        n_upd.append(oo.update(ot, cloud, with_color=color, threads=update_threads))
        t2 = time.perf_counter()
        t_track += t1 - t0
        t_upd += t2 - t1
        iters.append(so["iterations"]); stops.append(bool(so["stopped"])); poses.append(ot.trans.copy()); rots.append(ot.rot.copy())
        done += 1
        if time.perf_counter() - t_all > time_limit_s:
            break
    timing = {"frames": done, "track_s": t_track, "update_s": t_upd}
    # the same frames through the C ABI (host planes), twice:
    #   free-running   every pose the HIP tracker estimates feeds its next integration, as in production.  The poses drift
    #                  apart by ~1e-13 m (another summation order of the normal equations), which now and then flips the f32
    #                  rounding of a voxel's distance: D may differ by an ulp of that distance ANYWHERE (bar: SURVEY 8c's
    #                  2e-6 m), W and the colour weights do not depend on the distance's last bit.
    #   teacher-forced every frame is tracked from the ORACLE's previous pose and integrated at the ORACLE's new pose: same
    #                  inputs to every kernel call, so every array must be bit-identical except in voxels that took the
    #                  exp() weight (the oracle records them), and every tracked pose within 1e-9 of the oracle's.
    mask = oo.exp_mask

    def differing(got, want):
        idx = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        if idx.size:                                     # +0 / -0 and NaN payloads are not differences
            a, b = got[idx], want[idx]
            idx = idx[~(((a == 0) & (b == 0)) | (np.isnan(a) & np.isnan(b)))]
        if not idx.size:
            return {"voxels": 0, "max_ulp": 0, "max_abs": 0.0, "outside_the_exp_band": 0}
        a, b = got[idx], want[idx]
        ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
        ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
        return {"voxels": int(idx.size), "max_ulp": int(np.max(np.abs(ia - ib))),
                "max_abs": float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))),
                "outside_the_exp_band": int((mask[idx] == 0).sum())}

    def compare_volume(gs):
        D, W = gs.download()
        arrays = {"W": differing(W, oo.W), "D": differing(D, oo.D)}
        del D, W
        if color:
            for name, got, want in zip(("Color_W", "R", "G", "B"), gs.download_color(), (oo.Color_W, oo.R, oo.G, oo.B)):
                arrays[name] = differing(got, want)
        return arrays
    col = (lambda f: f) if color else (lambda f: (f[0], f[1], None))
    parity = {"frames": done + 1, "m": m, "image": [width, height], "carry_threads": track_threads,
              "voxels": int(oo.W.size), "voxels_that_took_the_exp_weight": int(mask.sum())}
    # ---- free-running
    gs = ts.SDF(m, with_color=color, device=dev_index, carry_threads=track_threads)
    gt = ts.CameraTracking(sdf=gs)
    gt.set_K(K)
    try:
        g_upd = [gs.update(gt, *col(frames[0]))["n_updated"]]
        g_iters, g_stops, gaps = [], [], []
        for k in range(1, done + 1):
            sg = gt.estimate_new_position(gs, frames[k][0])
            g_upd.append(gs.update(gt, *col(frames[k]))["n_updated"])
            g_iters.append(int(sg["iterations"])); g_stops.append(bool(sg["stopped"]))
            gaps.append(float(np.max(np.abs(gt.trans - poses[k]))))
        arrays = compare_volume(gs)
    finally:
        gs.close()
    n_vox = float(oo.W.size)
    free = {"iterations_equal": g_iters == iters, "stop_flags_equal": g_stops == stops, "n_updated_equal": g_upd == n_upd[:done + 1],
            "iteration_mismatches": int(sum(a != b for a, b in zip(g_iters, iters))),
            "n_updated_mismatches": int(sum(a != b for a, b in zip(g_upd, n_upd))),
            "max_pose_gap_m": max(gaps) if gaps else 0.0, "pose_gap_bar_m": 1e-5 if done <= 10 else 1e-4, "arrays": arrays,
            "bars": "iterations, stop flags, n_updated equal per frame; pose gap <= pose_gap_bar_m; W <= 1 ulp; D abs <= 2e-6 m (SURVEY 8c); "
                    "colour lanes <= 4 ulp; fewer than 1e-5 of the voxels differ in any array"}
    free["ok"] = bool(free["iterations_equal"] and free["stop_flags_equal"] and free["n_updated_equal"]
                      and free["max_pose_gap_m"] <= free["pose_gap_bar_m"] and arrays["W"]["max_ulp"] <= 1 and arrays["D"]["max_abs"] <= 2e-6
                      and all(arrays[k]["max_ulp"] <= 4 for k in arrays if k not in ("D", "W"))
                      and all(v["voxels"] / n_vox < 1e-5 for v in arrays.values()))
    parity["free_running"] = free
    # ---- teacher-forced (the oracle's run is replayed: its poses are the inputs)
    gs = ts.SDF(m, with_color=color, device=dev_index, carry_threads=track_threads)
    gt = ts.CameraTracking(sdf=gs)
    gt.set_K(K)
    try:
        t_upd = [gs.update(gt, *col(frames[0]))["n_updated"]]
        t_iters, t_gaps = [], []
        for k in range(1, done + 1):
            gt.set_camera_transformation(rots[k - 1], poses[k - 1])
            sg = gt.estimate_new_position(gs, frames[k][0])
            t_iters.append(int(sg["iterations"]))
            t_gaps.append(float(max(np.max(np.abs(gt.trans - poses[k])), np.max(np.abs(gt.rot - rots[k])))))
            gt.set_camera_transformation(rots[k], poses[k])
            t_upd.append(gs.update(gt, *col(frames[k]))["n_updated"])
        arrays_t = compare_volume(gs)
    finally:
        gs.close()
    forced = {"iterations_equal": t_iters == iters, "n_updated_equal": t_upd == n_upd[:done + 1], "max_pose_gap_one_call": max(t_gaps) if t_gaps else 0.0,
              "pose_gap_bar": 1e-9, "arrays": arrays_t,
              "differences_outside_the_exp_band": int(sum(v["outside_the_exp_band"] for v in arrays_t.values())),
              "bars": "every estimate_new_position from the oracle's pose: same iterations, pose and rotation <= 1e-9; every update at the oracle's "
                      "pose: n_updated equal, every array bit-identical except in voxels that took the exp() weight (W <= 1 ulp, D and colour <= 4 ulp there)"}
    forced["ok"] = bool(forced["iterations_equal"] and forced["n_updated_equal"] and forced["max_pose_gap_one_call"] <= 1e-9
                        and forced["differences_outside_the_exp_band"] == 0 and arrays_t["W"]["max_ulp"] <= 1
                        and all(v["max_ulp"] <= 4 for v in arrays_t.values()))
    parity["teacher_forced"] = forced
    parity["ok"] = bool(free["ok"] and forced["ok"])
    # (the flat fields round 5's line carried, from the free-running run)
    parity.update({"iterations_equal": free["iterations_equal"], "stop_flags_equal": free["stop_flags_equal"], "n_updated_equal": free["n_updated_equal"],
                   "max_pose_gap_m": free["max_pose_gap_m"], "voxels_with_other_W_bits": arrays["W"]["voxels"], "max_W_ulp": arrays["W"]["max_ulp"],
                   "voxels_with_other_D_bits": arrays["D"]["voxels"], "max_D_ulp": arrays["D"]["max_ulp"], "max_abs_D_m": arrays["D"]["max_abs"]})
    return timing, parity


def cpu_baseline(args, K, frames, m, width, height, ts=None, dev_index=0):
    """The reference's CPU path (the oracle restatement: same loop structure, 24 B/voxel
    global_coords table, AoS clouds, OpenMP) timed on this box's host cores on a bounded sample.
    With `ts` (the HIP binding) the SAME frames also go through the HIP path at the SAME size, free-running, and the two runs
    are compared frame by frame and voxel by voxel (free_run_compare): `parity_full_size` at the oracle's timed thread count,
    `parity_canonical_order` with one tracker thread (SURVEY 8a8': the canonical carry order), `parity_config2` at 256^3."""
    import oracle as orc
    cores = usable_cores()
    n = max(1, min(args.cpu_baseline_frames, len(frames) - 1))
    timing, parity = free_run_compare(args, ts, orc, K, frames, m, width, height, n, cores, cores, dev_index, True)
    done = timing["frames"]
    total = timing["track_s"] + timing["update_s"]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"value": done / total, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{done} frames (track + update) of the same {width}x{height} stream at "
                     f"{m}^3 after 1 fused frame; OpenMP on all {cores} host threads",
           "track_ms_per_frame": 1e3 * timing["track_s"] / done, "update_ms_per_frame": 1e3 * timing["update_s"] / done,
           "cpu_model": model}
    more = {}
    try:
        # the canonical order: ONE tracker thread (the carry never resets inside a pass); the update keeps all cores -- its
        # result does not depend on the thread count
        _, more["parity_canonical_order"] = free_run_compare(args, ts, orc, K, frames, m, width, height, min(8, n), 1, cores, dev_index, False,
                                                             time_limit_s=15.0)
        if m != 256:       # BASELINE config 2's size on the same stream
            _, more["parity_config2"] = free_run_compare(args, ts, orc, K, frames, 256, width, height, n, cores, cores, dev_index, False,
                                                         time_limit_s=15.0)
    except Exception as e:      # noqa: BLE001  (an extra comparison must not take the result line with it)
        more["parity_extra_error"] = f"{type(e).__name__}: {e}"
    return out, parity, more


PMC_GROUPS = (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum"), ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))
PMC_CORRECTION_NOTE = ("bytes from the L2's memory-side request counters by request size: read = 32 n32 + 128 n128 + 64 (RDREQ - n32 - n128), "
                       "write = 64 n64 + 32 (WRREQ - n64).  Calibrated on this chip with known-byte kernels in this kernel's access "
                       "widths (tools/pmc_calibrate.py, profiles/r02_fetch_calibration.json): the derived bytes equal the known bytes "
                       "(ratio 1.000 for 8 B and 16 B per lane, plain and non-temporal, reads and writes), while FETCH_SIZE itself "
                       "reports exactly 1/2 of them (every streaming read leaves L2 as a 128-byte request tallied as 64 B)")


def pmc_traffic(args, wl_args):
    """HBM bytes of one integrate launch (list_rows_kernel + integrate_kernel) of THIS workload, measured now:
    two child runs of this script under rocprofv3 --pmc, one counter group each (read requests by size, write
    requests by size; never combined with a trace domain), program directly after `--`.  Returns a dict or a reason."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 not found"
    res = {}
    work = tempfile.mkdtemp(prefix="tsdf_pmc_", dir="/tmp")
    try:
        for gi, group in enumerate(PMC_GROUPS):
            d = os.path.join(work, "g%d" % gi)
            cmd = ["rocprofv3", "--pmc"] + list(group) + ["--kernel-include-regex", "tsdf::(integrate|list_rows)", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--steps", "12", "--warmup", "2"] + wl_args
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            except subprocess.TimeoutExpired:
                return f"rocprofv3 --pmc {group[0]} pass timed out"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return f"rocprofv3 --pmc {group[0]} pass failed (rc {p.returncode}): {p.stderr[-300:]}"
            acc = {}
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] not in group:
                        continue
                    name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
                    a = acc.setdefault((name, r["Counter_Name"]), [0, 0.0])
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
            for (name, counter), (cnt, tot) in acc.items():
                res.setdefault(name, {})[counter] = tot / cnt
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out = {}
    for name, c in res.items():
        n, n32, n128 = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        w, w64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        out[name] = {"read_bytes": 32.0 * n32 + 128.0 * n128 + 64.0 * (n - n32 - n128), "write_bytes": 64.0 * w64 + 32.0 * (w - w64),
                     "requests": c}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run(args)


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import torch.distributed as dist
    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py --gpus {world}: only {ndev} GPU(s) visible (one rank per GPU; --dist-backend gloo "
                         f"lets ranks share a GPU for testing)")
    dev_index = local_rank if args.dist_backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
    cpu_or_dev = dev if args.dist_backend == "nccl" else "cpu"

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=cpu_or_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # ---- workload
    def resolve(config, m_override=None, w_override=None, h_override=None):
        m_ref, n_ref, w, h, label = WORKLOADS[config]
        m = m_ref if n_ref == 1 else int(round(m_ref * (world / n_ref) ** (1.0 / 3.0) / 2.0)) * 2
        return (m_override or m), (w_override or w), (h_override or h), label, ("strong" if n_ref == 1 else "weak")
    m, width, height, wl_label, scaling = resolve(args.config, args.m, args.width, args.height)
    noise = not args.no_noise

    def render_frames(w, h, n_frames, step=1, K=None, scene="plant"):
        """Synthetic input, identical on every rank, rendered on this rank's GPU and left there."""
        seq = synth.Sequence(n_frames=n_frames, width=w, height=h, noise=noise, holes=0.02 if noise else 0.0, step=step, K=K,
                             scene=scene)
        if len(seq) < n_frames:
            raise SystemExit(f"trajectory has only {len(seq)} poses, need {n_frames}")
        fr = [seq.frame_torch(k, dev) for k in range(n_frames)]
        torch.cuda.synchronize()
        return seq, fr

    import ctypes as C
    L = ts.lib()
    f_set, f_step, f_pose = L.tsdf_set_frame_device, L.tsdf_track_and_integrate, L.tsdf_get_pose
    f_queue, f_next = L.tsdf_queue_frame_device, L.tsdf_next_frame
    MAIN_MODE = "device_q" if args.frame_queue else "device"
    perf = time.perf_counter

    class Leg:
        """One volume (this rank's slab) + the frame loop of sdf_reconstruction.cpp:69-74 over a list of frames."""

        def __init__(self, m, w, h, K, path=None):
            self.m, self.w, self.h = m, w, h
            cfg0 = ts.default_config(m=m)
            self.halo = ts.halo_for(cfg0, args.max_range) if world > 1 else 0
            # The placements a run may use: {name: (x0, x1, stride, owned ranges of every rank)}.
            placements = {}
            if world > 1:
                cuts_u = [ts.slab_range(m, world, r)[0] for r in range(world)] + [m]
                placements["uniform"] = (cuts_u[rank], cuts_u[rank + 1], 0, [[(cuts_u[r], cuts_u[r + 1])] for r in range(world)])
                if path is not None:
                    # every rank computes the same cuts: the frusta of the poses this run will visit, up to the sensor's 5 m;
                    # boundaries that minimise the path-average of the busiest rank (tracking_sdf_amd.slab_cuts_for_path)
                    cuts = ts.slab_cuts_for_path(cfg0, K, w, h, path[0], path[1], world, self.halo, 5.0)
                    placements["path"] = (cuts[rank], cuts[rank + 1], 0, [[(cuts[r], cuts[r + 1])] for r in range(world)])
                if args.slabs == "balanced":
                    wts = ts.frustum_layer_weights(cfg0, K, w, h, [[1, 0, 0], [0, 0, -1], [0, -1, 0]], [0, 0, 1], 5.0)
                    rb = [ts.slab_range_weighted(m, world, r, self.halo, wts) for r in range(world)]
                    placements["balanced"] = (rb[rank][0], rb[rank][1], 0, [[tuple(x)] for x in rb])
                # block-cyclic placement: two blocks per rank unless told otherwise; blocks must be wider than the halo allows
                # (stride = N B >= B + 2 halo) and divide the power-of-two m
                try:
                    cx0, cx1, cst = ts.cyclic_range(m, world, rank, self.halo, args.cyclic_block)      # tsdf_cyclic_range
                    blk = cx1 - cx0
                    placements["cyclic"] = (cx0, cx1, cst, [[(a, a + blk) for a in range(r * blk, m, cst)] for r in range(world)])
                except ts.TsdfError:
                    pass
                if args.slabs == "cyclic" and "cyclic" not in placements:
                    raise SystemExit(f"--slabs cyclic: no block size fits m={m}, {world} ranks, halo {self.halo}")
            # A frame waits for its busiest rank (the all-reduce of every pass synchronises them): what a placement costs is the
            # busiest rank's share of THAT frame's frustum work (the layers it STORES, halo included), averaged over the frames
            # of the planned path -- DESIGN 6.1's model, also printed in the result line (1 for a single rank).
            prefix_sums = []              # of the frustum weights of (up to 24 of) the path's poses, made once per leg
            if world > 1 and path is not None:
                for k in np.unique(np.linspace(0, len(path[0]) - 1, min(24, len(path[0]))).astype(int)):
                    pre = np.concatenate([[0.0], np.cumsum(ts.frustum_layer_weights(cfg0, K, w, h, path[0][k], path[1][k], 5.0))])
                    if pre[-1] > 0:
                        prefix_sums.append(pre)

            def busiest_share(owned):
                shares = [max(sum(pre[min(m, b + self.halo)] - pre[max(0, a - self.halo)] for a, b in owned[r]) for r in range(world)) / pre[-1]
                          for pre in prefix_sums]
                return float(np.mean(shares)) if shares else 1.0
            self.busiest_share, self.placement_shares = 1.0, None
            x0, x1, stride, policy = 0, m, 0, None
            if world > 1:
                if args.slabs == "auto":
                    # the path is known (this run's own): take the placement the model prices lowest over it -- cuts tuned to a short
                    # window of the path beat the block-cyclic placement's halo work, a camera that sweeps across x does not
                    cand = [p_ for p_ in ("cyclic", "path") if p_ in placements] if path is not None else []
                    self.placement_shares = {p_: busiest_share(placements[p_][3]) for p_ in cand}
                    policy = min(cand, key=lambda p_: self.placement_shares[p_]) if cand else "uniform"
                else:
                    policy = args.slabs if args.slabs in placements else "uniform"
                x0, x1, stride, owned = placements[policy]
                self.busiest_share = self.placement_shares[policy] if self.placement_shares and policy in self.placement_shares else busiest_share(owned)
            self.slab_policy = policy
            self.slab = (x0, x1)
            self.slab_stride = stride
            self.sdf = ts.SDF(m, with_color=not args.no_color, slab=(x0, x1), halo=self.halo, slab_stride=stride, device=dev_index)
            self.trk = ts.CameraTracking(sdf=self.sdf)
            self.trk.set_K(K)
            self.pose_t = np.zeros(3)
            self.pose_ptr = self.pose_t.ctypes.data_as(C.POINTER(C.c_double))
            self.track_wall = 0.0
            self.est = []
            self.sdf_has_queued = False

        def first_frame(self, fr, mode=MAIN_MODE, host=None, depth16=None):
            """frame 1: integrate only at the reference's initial pose (sdf_reconstruction.cpp:69)"""
            self.est = []
            if mode in ("ref_calls", "ref_calls_r4", "ref_calls_normals"):
                self.sdf.update_aos(*host[0]) if mode != "ref_calls_r4" else (self.sdf.set_frame_aos(*host[0]), self.sdf.update(want_stats=False))
            else:
                self.feed(0, fr, mode, host, depth16)
                self.sdf.update(want_stats=False)
            self.est.append(self.trk.trans.copy())

        def feed(self, k, fr, mode, host, depth16):
            if mode == "device":
                dx, dn, dc = fr[k]
                self.sdf._check(f_set(self.sdf._h, C.c_void_p(dx.data_ptr()), C.c_void_p(dn.data_ptr()),
                                      C.c_void_p(dc.data_ptr()), self.w, self.h))
            elif mode == "device_q":
                # the two-deep queue with frames that are resident in HBM: frame k was queued during step k-1, frame k+1 is
                # queued now, so its packing kernel runs next to frame k's tracker passes instead of in front of frame k+1's
                def q(i):
                    dx, dn, dc = fr[i]
                    self.sdf._check(f_queue(self.sdf._h, C.c_void_p(dx.data_ptr()), C.c_void_p(dn.data_ptr()),
                                            C.c_void_p(dc.data_ptr()), self.w, self.h))
                if not self.sdf_has_queued:
                    q(k)
                self.sdf._check(f_next(self.sdf._h))
                self.sdf_has_queued = k + 1 < len(fr)
                if self.sdf_has_queued:
                    q(k + 1)
            elif mode == "caller_copies":
                # What a caller with page-locked frames can do with the DEVICE-frame entry point today: copy frames k+1 and k+2
                # itself, on a stream of its own, into a ring of device buffers while frame k is processed, and hand each over with
                # tsdf_set_frame_device once its copy is complete (host = the pinned tensors).  A ring slot is reused only after
                # tsdf_device_frame_released() says the library has packed the frame that was in it.
                st = getattr(self, "_cc", None)
                if st is None or st["host"] is not host:
                    n_slot = 4
                    st = self._cc = {"host": host, "stream": torch.cuda.Stream(device=dev), "ev": [torch.cuda.Event() for _ in range(n_slot)],
                                     "ring": [tuple(torch.empty_like(t, device=dev) for t in host[0]) for _ in range(n_slot)],
                                     "serial": [0] * n_slot, "issued": -1, "n": n_slot}
                def issue(i):
                    slot = i % st["n"]
                    while st["serial"][slot] and L.tsdf_device_frame_released(self.sdf._h) < st["serial"][slot]:
                        pass
                    with torch.cuda.stream(st["stream"]):
                        for dst, src in zip(st["ring"][slot], host[i]):
                            dst.copy_(src, non_blocking=True)
                        st["ev"][slot].record(st["stream"])
                    st["issued"] = i
                if k == 0:
                    st["issued"] = -1
                    st["serial"] = [0] * st["n"]
                for i in range(st["issued"] + 1, min(k + 3, len(host))):
                    issue(i)
                slot = k % st["n"]
                st["ev"][slot].synchronize()
                dx, dn, dc = st["ring"][slot]
                self.sdf._check(f_set(self.sdf._h, C.c_void_p(dx.data_ptr()), C.c_void_p(dn.data_ptr()), C.c_void_p(dc.data_ptr()), self.w, self.h))
                st["serial"][slot] = int(L.tsdf_frame_serial(self.sdf._h))
            elif mode == "host":
                self.sdf.set_frame(*host[k])
            elif mode == "aos":
                self.sdf.set_frame_aos(*host[k])
            elif mode == "depth_q":
                qd = lambda i: self.sdf.queue_depth_frame(depth16[i], host[i][2])
                if k == 0:
                    for j in range(min(args.queue_ahead, len(host))):
                        qd(j)
                self.sdf.next_frame()
                if k + args.queue_ahead < len(host):
                    qd(k + args.queue_ahead)
            elif mode in ("host_q", "aos_q"):
                # the frame queue: frame k was queued during step k-2 (k-1 with --queue-ahead 1) and becomes current now;
                # frame k+2 is queued before frame k is tracked and integrated, so its staging and upload have two frames' time
                q = self.sdf.queue_frame if mode == "host_q" else self.sdf.queue_frame_aos
                if k == 0:
                    for j in range(min(args.queue_ahead, len(host))):
                        q(*host[j])
                self.sdf.next_frame()
                if k + args.queue_ahead < len(host):
                    q(*host[k + args.queue_ahead])
            else:
                self.sdf.set_depth_frame(depth16[k], host[k][2])

        def step(self, k, fr, mode=MAIN_MODE, host=None, depth16=None, timed=False):
            if mode in ("ref_calls", "ref_calls_r4", "ref_calls_normals"):
                # exactly what kinect_callback does (sdf_reconstruction.cpp:70,74), synchronously, on pageable PCL-layout
                # clouds: estimate_new_position(sdf, cloud) then update(tracker, cloud, normals)
                pts, nn = host[k]
                # (raw ctypes calls with the pointers and the layout prepared once: the Python wrappers' per-call checks and
                # result dictionaries cost this loop ~10 % and are not part of what is measured)
                if not hasattr(self, "_aos_args") or self._aos_args[0] is not host:
                    lay = self.sdf._aos_layout(host[0][0], host[0][1])
                    self._aos_args = (host, lay, [(C.c_void_p(p_.ctypes.data), C.c_void_p(n_.ctypes.data)) for p_, n_ in host])
                lay, (pp, np_) = self._aos_args[1], self._aos_args[2][k]
                hh = self.sdf._h
                if mode == "ref_calls":
                    self.sdf._check(L.tsdf_track_aos(hh, pp, C.byref(lay), self.w, self.h, None))             # samples first, the cloud staged under the passes
                    self.sdf._check(L.tsdf_integrate_aos(hh, pp, np_, C.byref(lay), self.w, self.h, None))    # normals; the cloud is compared, not uploaded again
                elif mode == "ref_calls_normals":                  # one more argument at :70: the normals at tracking time
                    self.sdf._check(L.tsdf_track_frame_aos(hh, pp, np_, C.byref(lay), self.w, self.h, None))  # the whole frame staged and packed under the passes
                    self.sdf._check(L.tsdf_integrate(hh, None))
                else:                                              # round 4's shim: two uploads, a host wait for the frame stream
                    self.sdf.set_frame_aos(pts, None)
                    self.sdf._check(L.tsdf_track(self.sdf._h, None))
                    self.sdf.set_frame_aos(None, nn)
                    self.sdf._check(L.tsdf_integrate(self.sdf._h, None))
                f_pose(self.sdf._h, None, self.pose_ptr, None, None)
                self.est.append(self.pose_t.copy())
                return
            self.feed(k, fr, mode, host, depth16)
            tq = perf()
            rc = f_step(self.sdf._h, 1, None, None)        # estimate_new_position + update, sdf_reconstruction.cpp:70,74
            if timed:
                self.track_wall += perf() - tq
            self.sdf._check(rc)
            f_pose(self.sdf._h, None, self.pose_ptr, None, None)      # host-side pose read while the integration runs
            self.est.append(self.pose_t.copy())

        def timed_region(self, fr, mode=MAIN_MODE, host=None, depth16=None, events=True):
            """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks."""
            self.first_frame(fr, mode, host, depth16)
            for k in range(1, 1 + args.warmup):
                self.step(k, fr, mode, host, depth16)
            self.sdf.synchronize()
            # HIP events (on the library's own stream) around every n-th integrate / pack launch of the timed region:
            # an event pair around every launch costs the loop ~6 % of its rate; the tracker is wall-timed
            self.sdf.set_timing(events, track=False, period=args.timing_period)
            self.sdf.read_timing(reset=True)
            self.sdf.read_counters(reset=True)
            self.track_wall = 0.0
            barrier()
            torch.cuda.synchronize()
            t0 = perf()
            for k in range(1 + args.warmup, 1 + args.warmup + args.steps):
                self.step(k, fr, mode, host, depth16, True)
            self.sdf.synchronize()
            torch.cuda.synchronize()
            barrier()
            elapsed = max_over_ranks(perf() - t0)
            tm, cn = self.sdf.read_timing(), self.sdf.read_counters()
            self.sdf.set_timing(False)
            return elapsed, tm, cn

        def restart(self):
            self.sdf.reset()                                 # constructor state: volume and the reference's initial pose

        def close(self):
            self.sdf.close()

    n_frames = 1 + args.warmup + args.steps
    seq, d_frames = render_frames(width, height, n_frames, args.frame_step,
                                  np.array(FR3_K) if args.config == 4 and (width, height) == (640, 480) else None)
    leg = Leg(m, width, height, seq.K, (seq.R[:n_frames], seq.t[:n_frames]))
    leg_slab0 = leg.slab
    leg_slab_policy = leg.slab_policy
    leg_slab_stride = leg.slab_stride
    leg_placement_shares = leg.placement_shares
    leg_busiest_share = leg.busiest_share
    sdf = leg.sdf
    halo_main = leg.halo

    # ---- exchange step of a Gauss-Newton pass (N > 1)
    allreduce_kind = "none"
    exchange_us = {}
    exchange_trial = {}
    comm_state = {"kind": "none"}

    def all_agree(flag):
        t = torch.tensor([1 if flag else 0], device=cpu_or_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def probe(s, n=40):
        """Self-test + time of one exchange step (sum of 30 doubles over the ranks), microseconds."""
        try:
            got = s.allreduce(np.full(30, float(rank + 1)))
            good = bool(np.all(got == world * (world + 1) / 2))
        except Exception as e:      # noqa: BLE001  (an exchange step that fails here must not take the run down)
            print(f"[bench] rank {rank}: exchange self-test failed ({e})", file=sys.stderr)
            good = False
        if not all_agree(good):     # every rank takes the same branch: a rank that timed the step alone would wait for the others
            return False, float("inf")
        dist.barrier()
        t0 = perf()
        try:
            for _ in range(n):
                s.allreduce(np.ones(30))
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: exchange step failed while being timed ({e})", file=sys.stderr)
            good = False
        return good, 1e6 * (perf() - t0) / n

    def init_rccl(s):
        buf = C.create_string_buffer(128)
        if not all_agree(ts.lib().tsdf_comm_unique_id(buf) == 0):      # can every rank bind librccl at all?
            return False
        uid = torch.tensor(list(buf.raw), dtype=torch.uint8, device=cpu_or_dev)
        dist.broadcast(uid, 0)
        try:
            s.comm_init(world, rank, bytes(uid.cpu().tolist()))
            good = True
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: in-library RCCL failed ({e})", file=sys.stderr)
            good = False
        return all_agree(good)

    shm_serial = [0]

    def init_shm(s, peer=False):
        shm_serial[0] += 1
        names = [f"/tsdf_{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}_{shm_serial[0]}"]
        dist.broadcast_object_list(names, 0)
        try:
            (s.comm_init_peer if peer else s.comm_init_shm)(world, rank, names[0])
            good = True
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: {'device-side peer exchange' if peer else 'shared-memory fan-in'} failed ({e})", file=sys.stderr)
            good = False
        return all_agree(good)

    KIND = {"rccl": "rccl-in-library", "shm": "shared-memory fan-in", "peer": "device-side peer exchange"}

    def init_mode(s, mode):
        return init_rccl(s) if mode == "rccl" else init_shm(s, peer=(mode == "peer"))

    def torch_hook(s):
        scratch = torch.zeros(30, dtype=torch.float64, device=cpu_or_dev)

        def hook(arr):
            scratch.copy_(torch.from_numpy(arr))
            dist.all_reduce(scratch)
            arr[:] = scratch.cpu().numpy()
        s.set_allreduce_hook(hook)

    def setup_exchange(s, want):
        """Returns the kind in use.  `want`: rccl | shm | torch (a mode that fails its self-test falls through
        rccl -> shm -> torch, except when it was asked for by name on the command line)."""
        s.comm_finalize()
        s.set_allreduce_hook(None)
        order = {"rccl": ["rccl", "shm"], "shm": ["shm"], "peer": ["peer"], "torch": []}[want]
        if args.dist_backend != "nccl" and "rccl" in order and not args.rccl_under_gloo:
            order.remove("rccl")               # ranks may share a GPU under gloo: RCCL refuses that
        for mode in order:
            if init_mode(s, mode):
                good, _ = probe(s, 4)
                if all_agree(good):
                    return KIND[mode]
            s.comm_finalize()
            dist.barrier()
        torch_hook(s)
        return "torch.distributed-hook"

    if world > 1:
        # time the in-library exchange steps on this machine (reported either way)
        for mode in ("rccl", "shm", "peer"):
            if mode == "rccl" and args.dist_backend != "nccl" and not args.rccl_under_gloo:
                continue
            if init_mode(sdf, mode):
                good, us = probe(sdf)
                if all_agree(good):
                    exchange_us[mode] = us
            sdf.comm_finalize()
            dist.barrier()
        want = args.allreduce
        if want == "auto":
            # What an exchange step costs as a stand-alone tsdf_allreduce (above) includes a copy, a launch and a wait that a
            # tracker pass does not pay, differently for each of the three: the choice is made on a short trial of the frame
            # loop itself (first frame + up to 8 tracked frames, max over ranks), same frames, volume restarted each time.
            trial_n = min(8, len(d_frames) - 1)
            for mode in list(exchange_us):
                if setup_exchange(sdf, mode) != KIND[mode]:
                    continue
                good, t_trial = True, float("inf")
                try:                    # a step that fails on this node (a rank that does not show up ...) must not take the run down
                    leg.restart()
                    leg.first_frame(d_frames)
                    sdf.synchronize()
                    barrier()
                    t0 = perf()
                    for k in range(1, 1 + trial_n):
                        leg.step(k, d_frames)
                    sdf.synchronize()
                    t_trial = perf() - t0
                except Exception as e:      # noqa: BLE001
                    print(f"[bench] rank {rank}: exchange step '{mode}' failed in its trial ({e})", file=sys.stderr)
                    good = False
                if all_agree(good):         # every rank takes the same branch
                    exchange_trial[mode] = trial_n / max_over_ranks(t_trial)
            leg.restart()
            choice = [max(exchange_trial, key=exchange_trial.get) if exchange_trial else (min(exchange_us, key=exchange_us.get) if exchange_us else "torch")]
            dist.broadcast_object_list(choice, 0)           # every rank must take the same decision: rank 0's
            want = choice[0]
        allreduce_kind = setup_exchange(sdf, want)
        if args.allreduce in ("shm", "peer") and allreduce_kind != KIND[args.allreduce]:
            raise SystemExit(f"--allreduce {args.allreduce} requested but it failed its self-test")
        if args.allreduce == "rccl" and args.dist_backend == "nccl" and not allreduce_kind.startswith("rccl") and rank == 0:
            print(f"[bench] in-library RCCL failed its self-test; exchange step = {allreduce_kind}", file=sys.stderr)
        comm_state["kind"] = allreduce_kind
        dist.barrier()

    # ---- the timed region: frames resident in HBM
    elapsed, tm, cn = leg.timed_region(d_frames)
    est_main = np.array(leg.est)
    track_wall_main = leg.track_wall

    if args.pmc_child:                       # child of pmc_traffic(): the launches above are all that is wanted
        print(json.dumps({"pmc_child": True, "steps": args.steps}))
        sdf.close()
        return

    extras = {}
    n1_extras = world == 1 and not args.no_extras
    host_frames = None
    if n1_extras or (world == 1 and not args.no_cpu_baseline):
        host_frames = [(x.cpu().numpy(), n.cpu().numpy(), c.cpu().numpy()) for x, n, c in d_frames]
        pinned_keep = [tuple(t.cpu().pin_memory() for t in fr) for fr in d_frames] if n1_extras else []
        pinned_frames = [tuple(t.numpy() for t in fr) for fr in pinned_keep]

    # Everything below is extra to the timed region above: a leg that fails must not take the result line with it.
    # Every rank runs the same legs; after each one the ranks agree on whether all of them got through, and skip the
    # rest together otherwise (a rank alone in a collective would hang the job).
    state = {"leg": leg, "frames": d_frames, "kind": allreduce_kind, "go": True}

    def guarded(name, fn):
        if not state["go"]:
            return
        ok = True
        try:
            fn()
        except Exception as e:      # noqa: BLE001
            ok = False
            extras[name + "_error"] = f"{type(e).__name__}: {e}"
            print(f"[bench] rank {rank}: extra leg '{name}' failed: {e}", file=sys.stderr)
        if world > 1:
            ok = all_agree(ok)
        state["go"] = ok

    # ---- PCIe-inclusive rates (H2D of the images + track + integrate), same frames, volume restarted
    def leg_h2d():
        lg = state["leg"]

        def best_of_two(mode, host, depth=None, reps=2, keep=None):
            # host-side legs (a CPU memcpy per frame) are exposed to scheduler hiccups of the box (a timed region is 12 ms
            # long, the box a 16-CPU quota next to other tenants): two repetitions, the better one is reported; `keep`
            # (the end-to-end leg: three) receives every repetition's rate
            best = None
            for _ in range(reps):
                lg.restart()
                e, _, _ = lg.timed_region(d_frames, mode, host, depth, events=False)
                best = e if best is None else min(best, e)
                if keep is not None:
                    keep.append(args.steps / e)
            return best
        other = "device_q" if MAIN_MODE == "device" else "device"
        extras["value_device_resident_queued" if other == "device_q" else "value_device_resident_one_at_a_time"] = args.steps / best_of_two(other, None)
        if other == "device":
            extras["value_device_resident_queued"] = args.steps / elapsed
        extras["device_queued_note"] = ("HBM-resident frames through tsdf_queue_frame_device / tsdf_next_frame (`value` since round 6): frame k+1 is packed, "
                                        "sample list included, by workgroups appended to frame k's integrate launch; one at a time "
                                        "(tsdf_set_frame_device, `value` until round 5): the launch packs frame k itself and the first pass reads its "
                                        "samples from the xyz plane")
        e2 = best_of_two("host", host_frames)
        extras["value_h2d_inclusive"] = args.steps / e2
        extras["h2d_inclusive_note"] = ("best of two repetitions; xyz + normals + rgb (27 B/pixel) handed over as HOST buffers every frame through "
                                        "tsdf_set_frame, one frame at a time: the tracker's 34 240 samples are copied first and its passes run "
                                        "under the planes' copy (round 5); staging copy + H2D + pack on the frame side stream")
        extras["value_h2d_inclusive_caller_side_copies"] = args.steps / best_of_two("caller_copies", pinned_keep)
        extras["caller_side_copies_note"] = ("page-locked frames copied by the CALLER on a stream of its own, two frames ahead, into a ring of four device "
                                             "buffers, each handed over with tsdf_set_frame_device when its copy is complete and reused once "
                                             "tsdf_device_frame_released reports it packed: the PCIe copy is in the loop, the library sees device frames")
        e2p = best_of_two("host", pinned_frames)
        extras["value_h2d_inclusive_pinned_buffers"] = args.steps / e2p
        extras["h2d_inclusive_pinned_note"] = ("the same with the caller's buffers page-locked: tsdf_set_frame copies from them "
                                               "directly (no staging memcpy of 8.3 MB per frame on the host)")
        depth16 = [np.where(np.isnan(x[..., 2]), 0, np.round(x[..., 2] * 5000.0)).astype(np.uint16) for x, _, _ in host_frames]
        e3 = best_of_two("depth", host_frames, depth16)
        extras["value_depth_input_inclusive"] = args.steps / e3
        extras["depth_input_note"] = ("raw uint16 depth + rgb (5 B/pixel) as host buffers; back-projection, bilateral-grid filter "
                                      "and normals on the GPU (tsdf_set_depth_frame; PCL parity of that stage is unpinned)")
        # the frame as the reference's callback holds it: pcl::PointCloud<PointXYZRGB> + pcl::PointCloud<Normal>, 32-byte
        # structs in pageable memory (64 B/pixel read on the host by the library's staging threads)
        aos = []
        for x, n, c in host_frames:
            pts = np.zeros(x.shape[:2], dtype=ts.PCL_POINT_XYZRGB)
            pts["x"], pts["y"], pts["z"] = x[..., 0], x[..., 1], x[..., 2]
            pts["r"], pts["g"], pts["b"] = c[..., 0], c[..., 1], c[..., 2]
            nn = np.zeros(x.shape[:2], dtype=ts.PCL_NORMAL)
            nn["normal_x"], nn["normal_y"], nn["normal_z"] = n[..., 0], n[..., 1], n[..., 2]
            aos.append((pts, nn))
        # the rate through the reference's own two entry points, called the way kinect_callback calls them
        extras["value_reference_entry_points"] = args.steps / best_of_two("ref_calls", aos)
        extras["value_reference_entry_points_round4_sequence"] = args.steps / best_of_two("ref_calls_r4", aos)
        extras["value_reference_entry_points_normals_at_track"] = args.steps / best_of_two("ref_calls_normals", aos)
        extras["reference_entry_points_note"] = (
            "estimate_new_position(sdf, cloud) then update(tracker, cloud, normals), synchronously, one frame at a time, clouds as arrays of "
            "PCL's 32-byte structs in pageable memory (sdf_reconstruction.cpp:33-49,70,74) = tsdf_track_aos + tsdf_integrate_aos, which the "
            "exact-type shim forwards to: the tracker's 34 240 samples are copied first, the cloud is staged under the Gauss-Newton passes, "
            "update adds the normals and compares the cloud instead of uploading it again.  _round4_sequence: what the shim issued until "
            "round 4 (tsdf_set_frame_aos(points) -> tsdf_track -> tsdf_set_frame_aos(normals only) -> tsdf_integrate).  The normals are only "
            "handed over by update, so their repack and copy (3.7 MB) sit between a frame's last pass and its integration whatever the library does.  "
            "_normals_at_track: NOT the reference's call -- estimate_new_position(sdf, cloud, normals), one more argument at "
            "sdf_reconstruction.cpp:70 (kinect_callback holds the normals by then): the whole frame is staged and packed under the passes "
            "(tsdf_track_frame_aos) and update integrates what was staged")
        # ... and the same loop in C++ through the reference's exact signatures (tests/mock/refcall_demo.cpp, built against the
        # mock Eigen / PCL headers: test infrastructure; the library underneath is the product), on the same frames
        try:
            import struct
            import tempfile
            subprocess.check_call(["make", "-C", ROOT, "-s", "refcall_demo"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with tempfile.TemporaryDirectory(prefix="tsdf_refcall_", dir="/tmp") as td:
                fb = os.path.join(td, "frames.bin")
                with open(fb, "wb") as f:
                    f.write(struct.pack("<3i", len(host_frames), width, height))
                    f.write(np.ascontiguousarray(seq.K, dtype="<f8").tobytes())
                    for k, (x, n_, c) in enumerate(host_frames):
                        f.write(struct.pack("<d", float(seq.stamps[k])))
                        f.write(np.ascontiguousarray(x, dtype="<f4").tobytes()); f.write(np.ascontiguousarray(n_, dtype="<f4").tobytes())
                        f.write(np.ascontiguousarray(c, dtype=np.uint8).tobytes())
                for key, flag in (("value_reference_entry_points_cpp", []), ("value_reference_entry_points_normals_at_track_cpp", ["normals-at-track"])):
                    best = 0.0
                    for _ in range(2):
                        pr = subprocess.run([os.path.join(ROOT, "build", "refcall_demo"), fb, str(m), os.path.join(td, "traj.txt")] + flag,
                                            capture_output=True, text=True, timeout=300)
                        for line in pr.stderr.splitlines():
                            if line.startswith("RATE "):
                                best = max(best, float(line.split()[1]))
                    extras[key] = best if best > 0 else None
        except Exception as e:      # noqa: BLE001
            extras["value_reference_entry_points_cpp"] = None
            extras["reference_entry_points_cpp_error"] = f"{type(e).__name__}: {e}"
        e4 = best_of_two("aos", aos)
        extras["value_pcl_clouds_inclusive"] = args.steps / e4
        extras["pcl_clouds_note"] = ("frames handed over as arrays of PCL's 32-byte point / normal structs in pageable memory "
                                     "(tsdf_set_frame_aos): what the reference's callback holds, sdf_reconstruction.cpp:33-49")
        # the same three host-buffer workloads through the frame queue (tsdf_queue_frame / tsdf_next_frame, --queue-ahead frames waiting)
        extras["end_to_end_repetitions"] = []
        extras["value_h2d_inclusive_queued"] = args.steps / best_of_two("host_q", [tuple(np.ascontiguousarray(a) for a in f) for f in host_frames],
                                                                        reps=3, keep=extras["end_to_end_repetitions"])
        extras["value_h2d_inclusive_pinned_buffers_queued"] = args.steps / best_of_two("host_q", pinned_frames)
        extras["value_pcl_clouds_inclusive_queued"] = args.steps / best_of_two("aos_q", aos)
        # SURVEY 8(d)'s end-to-end definition in one place: the reference's own input format (PCL clouds in pageable memory)
        # and the best a host can do (page-locked planes), both through the queue.  `value` stays the device-resident rate
        # the bench contract prescribes ("inputs already resident in HBM when the timed region starts").
        extras["value_host_inclusive"] = {"pcl_clouds_pageable": extras["value_pcl_clouds_inclusive_queued"],
                                          "planes_page_locked": extras["value_h2d_inclusive_pinned_buffers_queued"],
                                          "planes_pageable": extras["value_h2d_inclusive_queued"], "unit": "frames/s"}
        extras["queued_note"] = ("frame k+1 handed to tsdf_queue_frame(_aos) before frame k is tracked and integrated, taken with tsdf_next_frame: "
                                 "the upload (and the host-side repack of pageable buffers, on library threads) overlaps the whole of frame k")
        pin16 = [torch.from_numpy(d.view(np.int16)).pin_memory() for d in depth16]
        e3p = best_of_two("depth", pinned_frames, [t.numpy().view(np.uint16) for t in pin16])
        extras["value_depth_input_inclusive_pinned_buffers"] = args.steps / e3p
        extras["value_depth_input_inclusive_queued"] = args.steps / best_of_two("depth_q", host_frames, depth16)
        extras["value_depth_input_inclusive_pinned_buffers_queued"] = args.steps / best_of_two(
            "depth_q", pinned_frames, [t.numpy().view(np.uint16) for t in pin16])
        extras["value_host_inclusive"]["raw_depth_pageable"] = extras["value_depth_input_inclusive_queued"]
        extras["depth_queued_note"] = ("raw depth + rgb through tsdf_queue_depth_frame: upload, pre-processing (with its host round trip for "
                                       "the bilateral grid's depth range) and packing of frame k+1 on a library thread + the frame stream "
                                       "while frame k is tracked and integrated")
    if n1_extras:
        guarded("h2d_inclusive", leg_h2d)

    # ---- N > 1: the same timed region with the other exchange step, for comparison
    def leg_other_exchange():
        # the same timed region with each of the other in-library exchange steps (the main line ran with `allreduce_kind`)
        lg = state["leg"]
        chosen = [k for k, v in KIND.items() if v == allreduce_kind]
        for mode, key in (("rccl", "value_with_in_library_rccl"), ("shm", "value_with_shared_memory_fan_in"),
                          ("peer", "value_with_device_side_peer_exchange")):
            if mode in exchange_us and mode not in chosen and setup_exchange(lg.sdf, mode) == KIND[mode]:
                lg.restart()
                e4, _, _ = lg.timed_region(d_frames, events=False)
                extras[key] = args.steps / e4
        state["kind"] = setup_exchange(lg.sdf, chosen[0] if chosen else "rccl")
        dist.barrier()
    if world > 1 and not args.no_extras and len(exchange_us) > 1:
        guarded("other_exchange", leg_other_exchange)

    # ---- full fr1/plant sequence (1246 frames): ATE-RMSE and tracking failures at 256^3 and at the benchmark m
    def leg_full_sequence():
        state["leg"].close()
        state["leg"] = None
        torch.cuda.empty_cache()
        extras["full_sequence"] = full_sequence(ts, synth, torch, dev, dev_index, sorted({256, m}), width, height, noise,
                                                not args.no_color)
        extras["ate_full_sequence_m"] = extras["full_sequence"][str(m)]["ate_rmse_m"]
    if n1_extras and not args.no_full_sequence and args.config in (2, 3) and args.frame_step == 1:
        guarded("full_sequence", leg_full_sequence)

    # ---- the integrate launch on ROUND 1's benchmark scene (an almost empty room: 72 % of the listed lanes updated against
    # 53 % here), the fixed yardstick for the kernel across rounds: round 1 measured 0.1475 ms per launch pair, 0.373 of 8 TB/s
    def leg_round1_scene():
        if state["leg"] is not None:
            state["leg"].close()
            state["leg"] = None
        torch.cuda.empty_cache()
        seq1, fr1 = render_frames(width, height, n_frames, args.frame_step, None, "room")
        leg1 = Leg(m, width, height, seq1.K)
        try:
            e1, tm1, cn1 = leg1.timed_region(fr1)
            l1 = max(1, cn1["integrate_calls"])
            ms1 = tm1["integrate_ms"] / max(1, tm1["integrate_launches"])
            bpv1 = 16 if args.no_color else 48
            upd1 = (cn1["n_updated"] + cn1["n_updated_halo"]) / l1
            alg1 = bpv1 * upd1 + width * height * (27 if not args.no_color else 24)
            extras["round1_scene"] = {"note": "same command on round 1's scene (synth scene 'room'); round 1: 0.1475 ms per launch, frac 0.373, 3943 frames/s",
                                      "value": args.steps / e1, "avg_launch_ms": ms1, "updated_voxels_per_launch": upd1,
                                      "work_items_per_launch": cn1["integrate_items"] / l1,
                                      "live_fraction_of_listed_lanes": upd1 / max(1.0, 64.0 * cn1["integrate_items"] / l1),
                                      "achieved": alg1 / (ms1 * 1e-3) / 1e9 if ms1 > 0 else 0.0,
                                      "frac": alg1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS if ms1 > 0 else 0.0,
                                      "gn_iterations_per_frame": cn1["track_iterations"] / max(1, cn1["track_calls"])}
        finally:
            leg1.close()
            torch.cuda.empty_cache()
    if n1_extras and args.config == 3 and not args.no_weak_leg:
        guarded("round1_scene", leg_round1_scene)

    # ---- a config-5-shaped leg (weak scaling: m = 2048 (N/8)^(1/3), 1280x960), a few frames
    def leg_weak():
        if state["leg"] is not None:
            state["leg"].close()
            state["leg"] = None
        state["frames"] = None
        torch.cuda.empty_cache()
        m5, w5, h5, label5, _ = resolve(5)
        keep = (args.steps, args.warmup)
        args.steps, args.warmup = min(keep[0], 12), min(keep[1], 2)
        leg5 = None
        try:
            seq5, fr5 = render_frames(w5, h5, 1 + args.warmup + args.steps, 1)
            leg5 = Leg(m5, w5, h5, seq5.K, (seq5.R[:len(fr5)], seq5.t[:len(fr5)]))
            if world > 1:
                k = state["kind"]
                setup_exchange(leg5.sdf, {v: k2 for k2, v in KIND.items()}.get(k, "torch"))
            e5, tm5, cn5 = leg5.timed_region(fr5)
            l5 = max(1, cn5["integrate_calls"])
            t5 = max(1, tm5["integrate_launches"])
            ms5 = tm5["integrate_ms"] / t5
            bpv5 = 16 if args.no_color else 48
            upd5 = (cn5["n_updated"] + cn5["n_updated_halo"]) / l5
            extras["weak_leg"] = {"workload": label5, "m": m5, "image": [w5, h5], "n_gpus": world, "scaling": "weak",
                                  "halo": leg5.halo, "steps": args.steps, "value": args.steps / e5, "unit": "frames/s",
                                  "ms_per_step": 1e3 * e5 / args.steps, "integrate_launch_ms_rank0": ms5,
                                  "updated_voxels_per_launch_rank0": upd5,
                                  "integrate_GBs_rank0": (bpv5 * upd5 + w5 * h5 * (27 if not args.no_color else 24)) / (ms5 * 1e-3) / 1e9 if ms5 > 0 else None,
                                  "gn_iterations_per_frame": cn5["track_iterations"] / max(1, cn5["track_calls"])}
        finally:
            args.steps, args.warmup = keep
            if leg5 is not None:
                leg5.close()
            torch.cuda.empty_cache()
    if not args.no_extras and not args.no_weak_leg and args.config == 3 and (world > 1 or n1_extras):
        d_frames = None
        guarded("weak_leg", leg_weak)
    leg = state["leg"]
    allreduce_kind = state["kind"]

    if rank == 0:
        gt = seq.t[:len(est_main)]
        ate = horn_rmse(est_main[1:], gt[1:])
        raw = float(np.sqrt(np.mean(np.sum((est_main[1:] - gt[1:]) ** 2, axis=1))))
        bpv = 16 if args.no_color else 48
        # algorithmic bytes of the integration = 16 B (48 B with colour) per UPDATED voxel + the frame's images
        # once, w*h*(12 + 12 + 3) B (24 without colour).  `frac` is priced on exactly that.  What this implementation moves on
        # top of it BY DESIGN is reported separately (`implementation_bytes_per_launch`): the images are re-packed into 32-byte
        # pixel records inside the launch (planes read once -- that read IS the 27 B/pixel above -- records written once and
        # read once by integrate_kernel).
        img_bytes = width * height * (27 if not args.no_color else 24)
        launches = max(1, cn["integrate_calls"])                   # all launches of the timed region (counters)
        timed = max(1, tm["integrate_launches"])                   # the ones bracketed by HIP events (every n-th)
        upd_per_launch = (cn["n_updated"] + cn["n_updated_halo"]) / launches
        # frames set with tsdf_set_frame_device are packed INSIDE the integrate launch (workgroups appended to
        # list_rows_kernel): the launch then also reads the three planes once and writes the records once
        pack_in_launch = tm["pack_launches"] == 0
        rec_b = 32 if not args.no_color else 24
        pack_bytes = width * height * 2 * rec_b if pack_in_launch else width * height * rec_b   # records written (in-launch packing) + read
        alg_bytes = bpv * upd_per_launch + img_bytes
        avg_ms = tm["integrate_ms"] / timed
        pack_ms = tm["pack_ms"] / max(1, tm["pack_launches"])
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": f"frames/sec (track + integrate per frame), synthetic fr1/plant stream, {m}^3 TSDF",
            "value": args.steps / elapsed, "value_device_resident": args.steps / elapsed, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32/f64",
            "dtype_note": "f32 voxel state and SDF samples, f64 geometry and normal equations (the reference's own mix)",
            "data": "synthetic (frames resident in HBM before the timed region" + ("; handed over one at a time with tsdf_set_frame_device" if not args.frame_queue else "; frame k+1 queued with "
                    "tsdf_queue_frame_device while frame k is tracked and integrated") + ")",
            "config": {"workload": f"{wl_label}: fr1/plant ground-truth camera path at 30 Hz (re-based to the reference's "
                                   f"initial pose), analytic scene (plant on a pedestal at the path's focus, room with "
                                   f"pillars/domes/furniture), {width}x{height} depth with Kinect noise + 2% holes, "
                                   f"{m}^3 voxels, 6x6x3.5 m volume, colour lanes {'off' if args.no_color else 'on'}; "
                                   f"TUM images are not available on the box",
                       "config": args.config, "m": m, "image": [width, height], "parallelism": f"x-slab x{world}" + ((f" (block-cyclic: rank r owns the layers [{leg_slab0[1] - leg_slab0[0]} r + {leg_slab_stride} j, ...) of every block j, {leg_slab0[1] - leg_slab0[0]} layers each)"
                                                                              if leg_slab_stride else f" ({leg_slab_policy} slabs, rank 0 owns layers [{leg_slab0[0]}, {leg_slab0[1]}))") if world > 1 else ""),
                       "halo": halo_main,
                       "allreduce": allreduce_kind, "exchange_step_us_measured": exchange_us,
                       "exchange_trial_frames_per_s": exchange_trial, "slabs": leg_slab_policy},
            "scaling_model": None if world == 1 else (lambda ppf, ex: {
                "predicted_value": 1e6 / (17.0 + (120.0 - 17.0) * leg_busiest_share + ppf * (26.0 + ex)),
                "formula": "frames/s = 1e6 / (17 + (120 - 17) * busiest_share + passes_per_frame * (26 + exchange_us)): DESIGN 6.1's model with the "
                           "single-GPU constants measured in rounds 5-6 (integrate launch 120 us of which 17 us do not shrink with the slab, a "
                           "tracker pass 26 us whatever the slab); busiest_share = the busiest rank's share of a frame's frustum work (the layers it "
                           "STORES, halo included), averaged over this run's path; exchange_us = this machine's measured time of the chosen exchange step as a stand-alone tsdf_allreduce (an "
                           "upper bound of what it costs inside a pass).  Printed next to the measured `value` so that the first run on a "
                           "multi-GPU node confirms or falsifies the model by itself.",
                "busiest_share": leg_busiest_share, "busiest_share_of_the_candidates": leg_placement_shares,
                "passes_per_frame": ppf, "exchange_us": ex})(
                    cn["track_iterations"] / max(1, cn["track_calls"]),
                    float(exchange_us.get({v: k for k, v in KIND.items()}.get(allreduce_kind, ""), 0.0))),
            "ate_rmse_m": ate, "ate_frames": len(est_main) - 1, "abs_trajectory_rmse_m": raw,
            "gn_iterations_per_frame": cn["track_iterations"] / max(1, cn["track_calls"]),
            "stage_ms_per_frame": {"track_wall": 1e3 * track_wall_main / args.steps,
                                   "integrate_launch": avg_ms, "pack_kernel": pack_ms,
                                   "pack_note": ("no launch of its own: frames set with tsdf_set_frame_device are packed by workgroups "
                                                 "appended to list_rows_kernel, inside the integrate launch (TSDF_DEFER_PACK=0: as before)")
                                                if tm["pack_launches"] == 0 else "pack_kernel, one launch per frame"},
            "roofline": {"kernel": "integrate (list_rows_kernel, whose appended workgroups also pack the frame's pixel records, + integrate_kernel: one launch of the two per frame; integrate_kernel is 88 % of the interval)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_updated_voxel": bpv,
                         "implementation_bytes_per_launch": alg_bytes + pack_bytes,
                         "frac_with_the_record_bytes": (alg_bytes + pack_bytes) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_ms > 0 else 0.0,
                         "algorithmic_bytes_note": f"SURVEY 8(d): updated voxels x {bpv} B + the frame's images once ({img_bytes} B = w*h*"
                                                   f"{27 if not args.no_color else 24}); the {rec_b}-byte pixel records this implementation "
                                                   f"writes and reads on top ({pack_bytes} B) are in implementation_bytes_per_launch / "
                                                   "frac_with_the_record_bytes, not in frac",
                         "updated_voxels_per_launch": upd_per_launch, "avg_launch_ms": avg_ms,
                         "timed_launches": tm["integrate_launches"], "launches": cn["integrate_calls"],
                         "work_items_per_launch": cn["integrate_items"] / launches,
                         "live_fraction_of_listed_lanes": upd_per_launch / max(1.0, 64.0 * cn["integrate_items"] / launches),
                         "sweep_equiv_GBs": bpv * cn["n_voxels_swept"] / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0},
            # the tsdf_track call also waits for the previous frame's integration (same stream), so a pass is priced
            # on what is left of the frame after the integrate and pack launches
            "tracker_gather": {"in_grid_samples_per_pass": cn["track_in_grid"] / max(1, cn["track_iterations"]),
                               "passes": cn["track_iterations"],
                               "passes_through_the_librarys_own_queue": cn.get("track_passes_own_queue"),
                               "avg_pass_wall_ms": max(0.0, 1e3 * elapsed - args.steps * (avg_ms + pack_ms)) / max(1, cn["track_iterations"]),
                               "track_call_wall_ms_incl_wait_for_integrate": 1e3 * track_wall_main / max(1, cn["track_iterations"]),
                               "achieved_GBs_on_832B_per_sample": 832.0 * cn["track_in_grid"]
                                   / max(1e-9, 1e-3 * max(0.0, 1e3 * elapsed - args.steps * (avg_ms + pack_ms))) / 1e9},
        }
        out.update(extras)
        if "value_h2d_inclusive_queued" in extras:
            # SURVEY 8(d)'s end-to-end rate ("H2D of images + track + integrate") next to `value`.  The bench contract of this
            # build fixes `value` to "inputs already resident in HBM when the timed region starts" (a PCIe-inclusive rate is
            # never `value`), VERDICT r5 asked for the opposite: both are first-class fields of the line, with their ratio.
            e2e = extras["value_h2d_inclusive_queued"]
            out["value_end_to_end"] = e2e
            out["end_to_end"] = {
                "value": e2e, "unit": "frames/s", "over_device_resident": e2e / (args.steps / elapsed),
                "what": "xyz + normals + rgb (27 B/pixel, 8.3 MB per 640x480 frame) handed over in PAGEABLE host memory every frame "
                        "through tsdf_queue_frame / tsdf_next_frame (%s while frame k is tracked and "
                        "integrated); same frames, same volume size, same kernels as `value`"
                        % ("frames k+1 and k+2 wait in the queue: k+2 is staged and copied" if args.queue_ahead == 2 else "frame k+1 is staged and copied"),
                "frames_waiting_in_the_queue": args.queue_ahead,
                "repetitions": extras.get("end_to_end_repetitions"),      # `value` here is the best of these
                "page_locked_planes": extras.get("value_h2d_inclusive_pinned_buffers_queued"),
                "pcl_clouds_pageable": extras.get("value_pcl_clouds_inclusive_queued"),
                "raw_depth_pageable": extras.get("value_depth_input_inclusive_queued"),
                "raw_depth_page_locked": extras.get("value_depth_input_inclusive_pinned_buffers_queued"),
                "one_frame_at_a_time_pageable": extras.get("value_h2d_inclusive"),
                "reference_entry_points_cpp": extras.get("value_reference_entry_points_cpp"),
                "caller_side_copies": extras.get("value_h2d_inclusive_caller_side_copies")}
        if leg is not None:
            leg.close()
            leg = None
        # measured ceiling for this access pattern (streaming 48 B/voxel RMW in 64-voxel items; tools/rmw_probe.hip)
        probe_f = os.path.join(ROOT, "profiles", "r01_rmw_probe.json")
        if os.path.exists(probe_f):
            with open(probe_f) as f:
                pj = json.load(f)
            ceil_gbs = pj["item_granular_rows_rmw_GBs"] if not args.no_color else pj["float2_rmw_GBs"]
            out["roofline"]["measured_rmw_ceiling_GBs"] = ceil_gbs
            out["roofline"]["frac_of_measured_ceiling"] = achieved / ceil_gbs
            out["roofline"]["ceiling_source"] = "profiles/r01_rmw_probe.json (build/rmw_probe on MI355X)"
        # HBM traffic of the integrate launch (its two kernels), measured now by two rocprofv3 --pmc child passes of this workload
        if n1_extras and not args.no_pmc:
            wl = ["--config", str(args.config), "--voxels", str(m), "--width", str(width), "--height", str(height),
                  "--frame-step", str(args.frame_step), "--timing-period", str(args.timing_period)]
            wl += (["--no-color"] if args.no_color else []) + (["--no-noise"] if args.no_noise else [])
            torch.cuda.empty_cache()
            got = pmc_traffic(args, wl)
            if isinstance(got, dict):
                def tot(prefix, field):
                    return sum(v[field] for k, v in got.items() if k.startswith(prefix))
                kernels3 = ("tsdf::integrate_kernel", "tsdf::list_rows_kernel")
                fetch = sum(tot(k, "read_bytes") for k in kernels3)
                write = sum(tot(k, "write_bytes") for k in kernels3)
                out["roofline"]["traffic"] = fetch + write
                out["roofline"]["traffic_detail"] = {
                    "read_bytes": fetch, "write_bytes": write,
                    "source": "two rocprofv3 --pmc child passes of this command's workload run by bench.py itself "
                              "(one counter group per pass, 12 timed steps), bytes per launch of the two kernels",
                    "correction": PMC_CORRECTION_NOTE, "ratio_to_algorithmic": (fetch + write) / alg_bytes,
                    "requests_integrate_kernel": {k: v["requests"] for k, v in got.items() if k.startswith("tsdf::integrate_kernel")}}
            else:
                out["roofline"]["traffic_detail"] = {"source": "unavailable: " + str(got)}
        if args.trajectory_out:
            with open(args.trajectory_out, "w") as f:
                for k in range(1, len(est_main)):
                    f.write("%.4f %.4f %.4f %.4f 0.0000 0.0000 0.0000 1.0000\n" % (seq.stamps[k], *est_main[k]))
        if world == 1 and not args.no_cpu_baseline:
            d_frames = None
            torch.cuda.empty_cache()
            out["cpu_baseline"], out["parity_full_size"], more = cpu_baseline(args, seq.K, host_frames, m, width, height, ts, dev_index)
            out.update(more)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def full_sequence(ts, synth, torch, dev, dev_index, ms, width, height, noise, with_color):
    """All 1246 frames of the fr1/plant path, rendered on the GPU one at a time and fed to one handle per m
    (frame 1 integrate only, then track -> integrate, sdf_reconstruction.cpp:69-74).  A tracking error leaves the
    pose where it was (tsdf_track's contract) and the frame is fused there, as tools/run_sequence.py does."""
    import ctypes as C
    seq = synth.Sequence(n_frames=None, width=width, height=height, noise=noise, holes=0.02 if noise else 0.0)
    n = len(seq)
    L = ts.lib()
    runs = []
    for m in ms:
        s = ts.SDF(m, with_color=with_color, device=dev_index)
        t = ts.CameraTracking(sdf=s)
        t.set_K(seq.K)
        runs.append({"m": m, "sdf": s, "trk": t, "est": np.zeros((n, 3)), "errors": 0, "first_error": None, "iters": 0})
    pose = np.zeros(3)
    pptr = pose.ctypes.data_as(C.POINTER(C.c_double))
    t0 = time.perf_counter()
    hot = 0.0
    for k in range(n):
        dx, dn, dc = seq.frame_torch(k, dev)
        torch.cuda.current_stream().synchronize()          # the library's stream borrows finished buffers
        th = time.perf_counter()
        for r in runs:
            s = r["sdf"]
            s._check(L.tsdf_set_frame_device(s._h, C.c_void_p(dx.data_ptr()), C.c_void_p(dn.data_ptr()),
                                             C.c_void_p(dc.data_ptr()), width, height))
            if k > 0:
                rc = L.tsdf_track(s._h, None)
                if rc != 0:
                    r["errors"] += 1
                    if r["first_error"] is None:
                        r["first_error"] = {"frame": k, "status": int(rc), "message": L.tsdf_last_error(s._h).decode()}
            s._check(L.tsdf_integrate(s._h, None))
            L.tsdf_get_pose(s._h, None, pptr, None, None)
            r["est"][k] = pose
        for r in runs:
            r["sdf"].synchronize()                         # the frame's buffers are reused by the next render
        hot += time.perf_counter() - th
    wall = time.perf_counter() - t0
    out = {"frames": n, "wall_s_incl_rendering": wall, "hot_path_s_all_volumes": hot}
    for r in runs:
        cn = r["sdf"].read_counters()
        est = r["est"]
        err = np.linalg.norm(est - seq.t, axis=1)
        out[str(r["m"])] = {"ate_rmse_m": horn_rmse(est[1:], seq.t[1:]), "abs_trajectory_rmse_m": float(np.sqrt(np.mean(err[1:] ** 2))),
                            "max_abs_error_m": float(err.max()), "track_errors": r["errors"], "first_error": r["first_error"],
                            "gn_iterations_per_frame": cn["track_iterations"] / max(1, cn["track_calls"])}
        r["sdf"].close()
    return out


if __name__ == "__main__":
    main()
