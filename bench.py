#!/usr/bin/env python3
"""bench.py -- frames/sec + ATE-RMSE of the tracking_sdf hot path on MI355X.

A "step" is one frame of the reference's per-frame hot path (sdf_reconstruction.cpp:69-74):
CameraTracking::estimate_new_position (<= 20 Gauss-Newton passes) followed by SDF::update, on
a 640x480 synthetic depth stream rendered along the real fr1/plant ground-truth camera path
(no TUM image data exists on the box), against a 512^3 TSDF with the reference's default volume.
All input frames are resident in HBM before the timed region starts.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; the volume is sharded into N
   x-slabs + halo, one 30-double RCCL all-reduce of the normal equations per Gauss-Newton pass.)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--voxels", dest="m", type=int, default=512, help="voxels per axis m (BASELINE metric: 512)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-color", action="store_true", help="drop the colour lanes (sdf.cpp:294-304)")
    ap.add_argument("--no-noise", action="store_true")
    ap.add_argument("--frame-step", type=int, default=1, help="use every n-th 30 Hz pose")
    ap.add_argument("--max-range", type=float, default=6.0, help="metres; sizes the slab halo")
    ap.add_argument("--cpu-baseline-frames", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allreduce", choices=["auto", "rccl", "shm", "torch"], default="auto",
                    help="auto = self-test and time in-library RCCL and the shared-memory fan-in, keep the faster (torch hook if both fail)")
    ap.add_argument("--host-frames", action="store_true", help="hand frames over as HOST buffers every step (PCIe-inclusive rate; not the headline value)")
    ap.add_argument("--depth-input", action="store_true", help="hand over raw uint16 depth + rgb as HOST buffers; back-projection, bilateral filter and normals run on the GPU (tsdf_set_depth_frame); PCIe-inclusive, not the headline value")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for the launcher plumbing (gloo lets several ranks share one GPU for testing)")
    ap.add_argument("--timing-period", type=int, default=4, help="HIP events around every n-th integrate/pack launch")
    ap.add_argument("--trajectory-out", default=None, help="write the estimated trajectory (TUM format)")
    return ap.parse_args()


def horn_rmse(est, gt):
    """ATE-RMSE after the rigid (rotation + translation) least-squares alignment of est onto gt."""
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


def cpu_baseline(args, seq, frames):
    """The reference's CPU path (the oracle restatement: same loop structure, 24 B/voxel
    global_coords table, AoS clouds, OpenMP) timed on this box's host cores on a bounded sample."""
    import oracle as orc
    cores = usable_cores()
    n = max(1, min(args.cpu_baseline_frames, len(frames) - 1))
    oo = orc.SDF(args.m, 6.0, 6.0, 3.5, (-3.0, -3.0, -0.5), 0.3, 0.025, with_global_coords=True)
    ot = orc.CameraTracking(oo)
    ot.set_K(seq.K)
    xyz, nrm, rgb = frames[0]
    oo.update(ot, orc.Cloud(xyz, nrm, rgb), with_color=not args.no_color, threads=cores)
    t_track = t_upd = 0.0
    done = 0
    t_all = time.perf_counter()
    for k in range(1, n + 1):
        xyz, nrm, rgb = frames[k]
        cloud = orc.Cloud(xyz, nrm, rgb)
        t0 = time.perf_counter()
        ot.estimate_new_position(oo, cloud, threads=cores, stale_carry=True)
        t1 = time.perf_counter()
        oo.update(ot, cloud, with_color=not args.no_color, threads=cores)
        t2 = time.perf_counter()
        t_track += t1 - t0
        t_upd += t2 - t1
        done += 1
        if time.perf_counter() - t_all > 20.0:
            break
    total = t_track + t_upd
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": done / total, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} frames (track + update) of the same {args.width}x{args.height} stream at "
                      f"{args.m}^3 after 1 fused frame; OpenMP on all {cores} host threads",
            "track_ms_per_frame": 1e3 * t_track / done, "update_ms_per_frame": 1e3 * t_upd / done,
            "cpu_model": model}


def main():
    args = parse()
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
    dev_index = local_rank if args.dist_backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- synthetic input (identical on every rank), uploaded to HBM before anything is timed
    n_frames = 1 + args.warmup + args.steps
    seq = synth.Sequence(n_frames=n_frames, width=args.width, height=args.height, noise=not args.no_noise,
                         holes=0.0 if args.no_noise else 0.02, step=args.frame_step)
    if len(seq) < n_frames:
        raise SystemExit(f"trajectory has only {len(seq)} poses, need {n_frames}")
    frames = [seq.frame(k) for k in range(n_frames)]
    d_frames = [(torch.from_numpy(x).to(dev), torch.from_numpy(n).to(dev), torch.from_numpy(c).to(dev))
                for x, n, c in frames]
    depth16 = [np.where(np.isnan(x[..., 2]), 0, np.round(x[..., 2] * 5000.0)).astype(np.uint16) for x, _, _ in frames] \
        if args.depth_input else None
    torch.cuda.synchronize()

    # ---- volume: x-slab of this rank (+ halo), colour lanes as in the reference
    x0, x1 = ts.slab_range(args.m, world, rank)
    cfg0 = ts.default_config(m=args.m)
    halo = ts.halo_for(cfg0, args.max_range) if world > 1 else 0
    sdf = ts.SDF(args.m, with_color=not args.no_color, slab=(x0, x1), halo=halo, device=dev_index)
    trk = ts.CameraTracking(sdf=sdf)
    trk.set_K(seq.K)

    allreduce_kind = "none"
    exchange_us = {}
    if world > 1:
        cpu_or_dev = dev if args.dist_backend == "nccl" else "cpu"

        def all_agree(flag):
            t = torch.tensor([1 if flag else 0], device=cpu_or_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        def probe(n=40):
            """Self-test + time of one exchange step (sum of 30 doubles over the ranks), microseconds."""
            got = sdf.allreduce(np.full(30, float(rank + 1)))
            good = bool(np.all(got == world * (world + 1) / 2))
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                sdf.allreduce(np.ones(30))
            return good, 1e6 * (time.perf_counter() - t0) / n

        def init_rccl():
            import ctypes
            buf = ctypes.create_string_buffer(128)
            if not all_agree(ts.lib().tsdf_comm_unique_id(buf) == 0):      # can every rank bind librccl at all?
                return False
            uid = torch.tensor(list(buf.raw), dtype=torch.uint8, device=cpu_or_dev)
            dist.broadcast(uid, 0)
            try:
                sdf.comm_init(world, rank, bytes(uid.cpu().tolist()))
                good = True
            except Exception as e:      # noqa: BLE001
                print(f"[bench] rank {rank}: in-library RCCL failed ({e})", file=sys.stderr)
                good = False
            return all_agree(good)

        def init_shm():
            names = [f"/tsdf_{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}"]
            dist.broadcast_object_list(names, 0)
            try:
                sdf.comm_init_shm(world, rank, names[0])
                good = True
            except Exception as e:      # noqa: BLE001
                print(f"[bench] rank {rank}: shared-memory fan-in failed ({e})", file=sys.stderr)
                good = False
            return all_agree(good)

        want = args.allreduce
        candidates = {"auto": ["rccl", "shm"], "rccl": ["rccl"], "shm": ["shm"], "torch": []}[want]
        if args.dist_backend != "nccl" and "rccl" in candidates:
            candidates.remove("rccl")          # ranks may share a GPU under gloo: RCCL refuses that
        for mode in candidates:
            if (init_rccl() if mode == "rccl" else init_shm()):
                good, us = probe()
                if all_agree(good):
                    exchange_us[mode] = us
            sdf.comm_finalize()
            dist.barrier()
        # every rank must take the same decision: use rank 0's timings
        choice = [min(exchange_us, key=exchange_us.get) if exchange_us else "torch"]
        dist.broadcast_object_list(choice, 0)
        if choice[0] == "rccl" and init_rccl():
            allreduce_kind = "rccl-in-library"
        elif choice[0] == "shm" and init_shm():
            allreduce_kind = "shared-memory fan-in"
        else:
            if want in ("rccl", "shm"):
                raise SystemExit(f"--allreduce {want} requested but it failed its self-test")
            sdf.comm_finalize()
            scratch = torch.zeros(30, dtype=torch.float64, device=cpu_or_dev)

            def hook(arr):
                scratch.copy_(torch.from_numpy(arr))
                dist.all_reduce(scratch)
                arr[:] = scratch.cpu().numpy()
            sdf.set_allreduce_hook(hook)
            allreduce_kind = "torch.distributed-hook"
        dist.barrier()

    track_wall = [0.0]
    est = []
    # The timed loop calls the three C-ABI entry points directly (pre-bound ctypes functions, pre-built pointer
    # arguments): the Python wrappers' dict/array conversions sit between tsdf_track returning and the integrate
    # launch, i.e. on the GPU's idle time.  Same calls, same error checks.
    import ctypes as C
    L = ts.lib()
    f_set, f_step, f_pose = L.tsdf_set_frame_device, L.tsdf_track_and_integrate, L.tsdf_get_pose
    handle = sdf._h
    dev_args = None
    if not (args.depth_input or args.host_frames):
        dev_args = [(C.c_void_p(dx.data_ptr()), C.c_void_p(dn.data_ptr()), C.c_void_p(dc.data_ptr())) for dx, dn, dc in d_frames]
    pose_t = np.zeros(3)
    pose_t_ptr = pose_t.ctypes.data_as(C.POINTER(C.c_double))
    perf = time.perf_counter

    def step(k, timed=False):
        if args.depth_input:
            sdf.set_depth_frame(depth16[k], frames[k][2])
        elif args.host_frames:
            sdf.set_frame(*frames[k])
        else:
            a = dev_args[k]
            sdf._check(f_set(handle, a[0], a[1], a[2], args.width, args.height))
        tq = perf()
        rc = f_step(handle, 1, None, None)                  # estimate_new_position + update, sdf_reconstruction.cpp:70,74
        if timed:
            track_wall[0] += perf() - tq
        sdf._check(rc)
        f_pose(handle, None, pose_t_ptr, None, None)        # host-side pose read while the integration runs
        est.append(pose_t.copy())

    # frame 0: integrate only at the reference's initial pose (sdf_reconstruction.cpp:69-74)
    if args.depth_input:
        sdf.set_depth_frame(depth16[0], frames[0][2])
    else:
        dx, dn, dc = d_frames[0]
        sdf.set_frame_device(dx.data_ptr(), dn.data_ptr(), dc.data_ptr(), args.width, args.height)
    sdf.update(want_stats=False)
    est.append(trk.trans.copy())
    for k in range(1, 1 + args.warmup):
        step(k)

    sdf.synchronize()
    # HIP events around every 4th integrate / pack launch of the timed region (an event pair around every launch costs
    # the loop ~6 % of its rate: measured 3800 vs 4050 frames/s); the tracker is wall-timed
    sdf.set_timing(True, track=False, period=args.timing_period)
    sdf.read_timing(reset=True)
    sdf.read_counters(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(1 + args.warmup, n_frames):
        step(k, True)
    sdf.synchronize()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tm = sdf.read_timing()
    cn = sdf.read_counters()
    sdf.set_timing(False)

    if rank == 0:
        est = np.array(est)
        gt = seq.t[:len(est)]
        ate = horn_rmse(est[1:], gt[1:])
        raw = float(np.sqrt(np.mean(np.sum((est[1:] - gt[1:]) ** 2, axis=1))))
        bpv = 16 if args.no_color else 48
        img_bytes = args.width * args.height * 32          # packed 32-byte pixel records read by the kernel
        launches = max(1, cn["integrate_calls"])                 # all launches of the timed region (counters)
        timed = max(1, tm["integrate_launches"])                 # the ones bracketed by HIP events (every n-th)
        upd_per_launch = (cn["n_updated"] + cn["n_updated_halo"]) / launches
        alg_bytes = bpv * upd_per_launch + img_bytes
        avg_ms = tm["integrate_ms"] / timed
        pack_ms = tm["pack_ms"] / max(1, tm["pack_launches"])
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": f"frames/sec (track + integrate per frame), synthetic fr1/plant stream, {args.m}^3 TSDF",
            "value": args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32/f64",
            "dtype_note": "f32 voxel state and SDF samples, f64 geometry and normal equations (the reference's own mix)",
            "data": "synthetic" + (" (frames handed over as host buffers: PCIe-inclusive)" if args.host_frames else "")
                    + (" (raw uint16 depth + rgb handed over as host buffers, pre-processed on the GPU: PCIe-inclusive)" if args.depth_input else ""),
            "config": {"workload": f"fr1/plant ground-truth camera path at 30 Hz (re-based to the reference's initial "
                                   f"pose), analytic room+sphere+boxes scene, {args.width}x{args.height} depth with "
                                   f"Kinect noise + 2% holes, {args.m}^3 voxels, 6x6x3.5 m volume, "
                                   f"colour lanes {'off' if args.no_color else 'on'}; TUM fr1/plant images are not "
                                   f"available on the box",
                       "m": args.m, "image": [args.width, args.height], "parallelism": f"x-slab x{world}",
                       "halo": halo, "allreduce": allreduce_kind, "exchange_step_us_measured": exchange_us},
            "ate_rmse_m": ate, "abs_trajectory_rmse_m": raw,
            "gn_iterations_per_frame": cn["track_iterations"] / max(1, cn["track_calls"]),
            "stage_ms_per_frame": {"track_wall": 1e3 * track_wall[0] / args.steps,
                                   "integrate_launch": avg_ms, "pack_kernel": pack_ms},
            "roofline": {"kernel": "integrate (clip_rows_kernel + integrate_kernel, one launch pair per frame)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_updated_voxel": bpv,
                         "updated_voxels_per_launch": upd_per_launch, "avg_launch_ms": avg_ms,
                         "timed_launches": tm["integrate_launches"], "launches": cn["integrate_calls"],
                         "sweep_equiv_GBs": bpv * cn["n_voxels_swept"] / launches / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0},
            # the tsdf_track call also waits for the previous frame's integration (same stream), so a pass is priced
            # on what is left of the frame after the integrate and pack launches
            "tracker_gather": {"in_grid_samples_per_pass": cn["track_in_grid"] / max(1, cn["track_iterations"]),
                               "passes": cn["track_iterations"],
                               "avg_pass_wall_ms": max(0.0, 1e3 * elapsed - args.steps * (avg_ms + pack_ms)) / max(1, cn["track_iterations"]),
                               "track_call_wall_ms_incl_wait_for_integrate": 1e3 * track_wall[0] / max(1, cn["track_iterations"]),
                               "achieved_GBs_on_832B_per_sample": 832.0 * cn["track_in_grid"]
                                   / max(1e-9, 1e-3 * max(0.0, 1e3 * elapsed - args.steps * (avg_ms + pack_ms))) / 1e9},
        }
        # HBM traffic of the integrate launch from the committed rocprofv3 --pmc passes (bench.py cannot collect
        # PMC counters itself); only meaningful for the default workload
        pmc = os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")
        if os.path.exists(pmc) and args.m == 512 and (args.width, args.height) == (640, 480) and not args.no_color \
                and world == 1:
            with open(pmc) as f:
                out["roofline"]["traffic"] = json.load(f)["integrate_launch_traffic_bytes"]
            out["roofline"]["traffic_source"] = "profiles/r01_final_pmc_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"
        # measured ceiling for this access pattern (streaming 48 B/voxel RMW in 64-voxel items; tools/rmw_probe.hip)
        probe = os.path.join(ROOT, "profiles", "r01_rmw_probe.json")
        if os.path.exists(probe):
            with open(probe) as f:
                pj = json.load(f)
            ceil_gbs = pj["item_granular_rows_rmw_GBs"] if not args.no_color else pj["float2_rmw_GBs"]
            out["roofline"]["measured_rmw_ceiling_GBs"] = ceil_gbs
            out["roofline"]["frac_of_measured_ceiling"] = achieved / ceil_gbs
            out["roofline"]["ceiling_source"] = "profiles/r01_rmw_probe.json (build/rmw_probe on MI355X)"
        if args.trajectory_out:
            with open(args.trajectory_out, "w") as f:
                for k in range(1, len(est)):
                    f.write("%.4f %.4f %.4f %.4f 0.0000 0.0000 0.0000 1.0000\n" % (seq.stamps[k], *est[k]))
        if world == 1 and not args.no_cpu_baseline:
            sdf.close()
            del d_frames
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args, seq, frames)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
