#!/usr/bin/env python3
"""What one rank of an N-way x-slab split costs, measured ALONE on the one GPU of the box.

For N in --ranks and every rank r of N: create the rank's handle (slab [x0, x1) of tsdf_slab_range + halo, exactly as
bench.py --gpus N does), fuse --frames frames at ground-truth poses (the reference's _useGroundTruth mode: every rank
integrates the same frames at the same poses, so its launches cost what they cost in the real job), then time tracker
passes at the last pose.  A rank alone on a GPU costs what it would cost on its own GPU of an N-GPU node -- the exchange
step between the ranks is NOT in these numbers; tsdf_allreduce timings of the three in-library exchange steps come from
bench.py (config.exchange_step_us_measured) and the model below takes them as parameters.

Model:   frame(N) = max_r integrate(N, r) + passes * (max_r pass(N, r) + exchange(N))
The ranks meet once per Gauss-Newton pass, so the slowest rank sets the pace of every pass and of the integration.

Prints one JSON document:  python tools/rank_costs.py --shape config3 --ranks 1 2 4 8 > profiles/r05_rank_costs_config3.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {      # name -> (m at 8 ranks, weak?, width, height)
    "config3": (512, False, 640, 480),          # the metric: fixed 512^3 volume split N ways (strong scaling)
    "config4": (1024, True, 640, 480),          # m = 1024 (N/8)^(1/3): a rank's voxel count stays constant (weak)
    "config5": (2048, True, 1280, 960),         # m = 2048 (N/8)^(1/3)
}
FR3_K = [[535.4, 0.0, 320.1], [0.0, 539.2, 247.6], [0.0, 0.0, 1.0]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="config3", choices=sorted(SHAPES))
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--frame-step", type=int, default=2)
    ap.add_argument("--passes", type=int, default=60)
    ap.add_argument("--max-range", type=float, default=6.0)
    ap.add_argument("--cyclic-block", type=int, default=0, help="--slabs cyclic: layers per block (0 = m / (2 N), two blocks per rank)")
    ap.add_argument("--slabs", default="uniform", choices=["uniform", "balanced", "path", "cyclic"],
                    help="uniform: tsdf_slab_range (equal thickness); balanced: tsdf_slab_range_weighted on the frustum weights of the "
                         "reference's initial pose; path: on the weights accumulated over the WHOLE fr1/plant path (what bench.py's "
                         "--slabs path does with the path it is going to run); cyclic: block-cyclic placement (tsdf_config::slab_stride), "
                         "bench.py's default where the volume allows it")
    ap.add_argument("--path-window", type=int, nargs=2, default=None, metavar=("FIRST", "LAST"),
                    help="--slabs path: cut for these poses of the path only (what bench.py does with the poses of its own run) instead of the whole path")
    ap.add_argument("--start-frame", type=int, default=0, help="where on the fr1/plant path the fused frames start (0, 480, 1000 ...)")
    ap.add_argument("--passes-per-frame", type=float, default=3.1, help="Gauss-Newton passes per frame of the bench stream (driver line)")
    ap.add_argument("--exchange-us", type=float, nargs="*", default=[0.0, 4.0, 10.0, 25.0],
                    help="exchange step per pass to evaluate the model at (us): 0 = none, ~4 = host fan-in through shared memory, "
                         "~10 = a device-side peer exchange over xGMI (unmeasured), ~25 = a small RCCL all-reduce (unmeasured)")
    args = ap.parse_args()

    import torch
    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth

    m8, weak, w, h = SHAPES[args.shape]
    dev = torch.device("cuda", 0)
    K = np.array(FR3_K) if args.shape == "config4" else None
    full = synth.Sequence(n_frames=None, width=w, height=h, noise=True, holes=0.02, K=K)           # the whole 1246-pose path
    n_need = args.start_frame + args.frames * args.frame_step
    if n_need > len(full):
        raise SystemExit(f"--start-frame {args.start_frame}: the path has {len(full)} poses")
    seq = full
    pick = [args.start_frame + k * args.frame_step for k in range(args.frames)]
    seq_R = [full.R[i] for i in pick]; seq_t = [full.t[i] for i in pick]
    d = [full.frame_torch(i, dev) for i in pick]
    torch.cuda.synchronize()
    out = {"what": __doc__.split("\n\n")[0], "shape": args.shape, "image": [w, h], "frames_fused": args.frames,
           "passes_timed": args.passes, "colour": True, "slabs": args.slabs, "start_frame": args.start_frame, "by_ranks": {}}
    for n in args.ranks:
        m = m8 if not weak else int(round(m8 * (n / 8.0) ** (1.0 / 3.0) / 2.0)) * 2
        halo = ts.halo_for(ts.default_config(m=m), args.max_range) if n > 1 else 0
        rows = []
        weights = None
        if args.slabs == "balanced" and n > 1:
            weights = ts.frustum_layer_weights(ts.default_config(m=m), seq.K, w, h, [[1, 0, 0], [0, 0, -1], [0, -1, 0]], [0, 0, 1])
        cuts = None
        if args.slabs == "path" and n > 1:
            wa, wb = args.path_window if args.path_window else (0, len(full))
            cuts = ts.slab_cuts_for_path(ts.default_config(m=m), seq.K, w, h, full.R[wa:wb], full.t[wa:wb], n, halo)
        blk = 0
        if args.slabs == "cyclic" and n > 1:
            c0, c1, _ = ts.cyclic_range(m, n, 0, halo, args.cyclic_block)          # tsdf_cyclic_range (raises when nothing fits)
            blk = c1 - c0
        for r in range(n):
            x0, x1 = (cuts[r], cuts[r + 1]) if cuts is not None else ts.slab_range(m, n, r) if weights is None else ts.slab_range_weighted(m, n, r, halo, weights)
            if blk:
                x0, x1 = r * blk, (r + 1) * blk
            sdf = ts.SDF(m, with_color=True, slab=(x0, x1), halo=halo, slab_stride=n * blk if blk else 0)
            trk = ts.CameraTracking(sdf=sdf)
            trk.set_K(seq.K)
            sdf.set_timing(True)
            for rep in range(2):                   # the second sweep gives warm numbers (band capacities, XCD shares)
                sdf.read_timing(reset=True)
                sdf.read_counters(reset=True)
                per_frame = []
                for k in range(args.frames):
                    trk.set_camera_transformation(seq_R[k], seq_t[k])
                    sdf.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), w, h)
                    t0 = time.perf_counter()
                    sdf.update(want_stats=False)
                    sdf.synchronize()
                    per_frame.append(time.perf_counter() - t0)
                wall = float(np.median(per_frame))
                tm, cn = sdf.read_timing(), sdf.read_counters()
            sdf.set_timing(False)
            k = args.frames - 1
            trk.set_camera_transformation(seq_R[k], seq_t[k] + np.array([0.004, -0.003, 0.002]))
            sdf.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), w, h)
            for _ in range(5):
                trk.accumulate()
            walls = []
            for _ in range(args.passes):
                t0 = time.perf_counter()
                A, b, st = trk.accumulate()
                walls.append(time.perf_counter() - t0)
            pass_wall = float(np.median(walls))        # (host hiccups of a shared box: the median, not the mean)
            rows.append({"rank": r, "slab": [x0, x1], "block_stride": n * blk if blk else 0,
                         "stored_layers": (len(range(x0, m, n * blk)) * (blk + 2 * halo)) if blk else min(m, x1 + halo) - max(0, x0 - halo),
                         "integrate_launch_us": 1e3 * tm["integrate_ms"] / max(1, tm["integrate_launches"]),
                         "integrate_call_to_completion_wall_us_median": 1e6 * wall,
                         "work_items_per_launch": cn["integrate_items"] / max(1, cn["integrate_calls"]),
                         "updated_voxels_per_launch": (cn["n_updated"] + cn["n_updated_halo"]) / max(1, cn["integrate_calls"]),
                         "updated_in_halo_fraction": cn["n_updated_halo"] / max(1, cn["n_updated"] + cn["n_updated_halo"]),
                         "pass_wall_us": 1e6 * pass_wall, "samples_in_own_slab": st["n_in_grid_owned"]})
            sdf.close()
            torch.cuda.empty_cache()
        integ = max(x["integrate_launch_us"] for x in rows)
        pw = max(x["pass_wall_us"] for x in rows)
        model = {}
        for ex in args.exchange_us:
            frame_us = integ + args.passes_per_frame * (pw + ex)
            model["exchange_%g_us" % ex] = {"frame_us": frame_us, "frames_per_s": 1e6 / frame_us}
        out["by_ranks"][str(n)] = {"m": m, "halo": halo, "ranks": rows, "max_integrate_launch_us": integ, "max_pass_wall_us": pw,
                                   "mean_integrate_launch_us": float(np.mean([x["integrate_launch_us"] for x in rows])),
                                   "model": model}
    out["model_note"] = ("frame(N) = max_r integrate(N, r) + passes_per_frame * (max_r pass(N, r) + exchange); passes_per_frame = %g; every rank "
                         "measured alone on the GPU; the exchange step is a PARAMETER here (see --exchange-us), not a measurement" % args.passes_per_frame)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
