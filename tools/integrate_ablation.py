#!/usr/bin/env python3
"""Timing experiments on integrate_kernel in ONE process: for every value of TSDF_DEBUG_INTEGRATE given on the command
line a fresh handle integrates the same frames at the ground-truth poses (fusion-only mode) and the average launch time
(list_rows_kernel + integrate_kernel between HIP events) is printed.  Needs a library built with
-DTSDF_INTEGRATE_DEBUG=1 for the bits to do anything (TSDF_HIP_LIB=build/variants/libtsdf_hip_dbg.so); the results of
most bits are garbage, only the time means something.  Prints one JSON line per (scene, debug value).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--frame-step", type=int, default=8)
    ap.add_argument("--scenes", default="plant,room")
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--debug", default="0", help="comma-separated TSDF_DEBUG_INTEGRATE values (sums of bits, a+b allowed)")
    ap.add_argument("--no-color", action="store_true")
    args = ap.parse_args()
    import torch
    import tracking_sdf_amd as ts
    from tracking_sdf_amd import synth

    dev = torch.device("cuda", 0)
    vals = [sum(int(x) for x in v.split("+")) for v in args.debug.split(",")]
    for scene in args.scenes.split(","):
        seq = synth.Sequence(n_frames=args.frames, width=640, height=480, noise=True, holes=0.02, step=args.frame_step,
                             scene=scene)
        d = [seq.frame_torch(k, dev) for k in range(args.frames)]
        torch.cuda.synchronize()
        for v in vals:
            os.environ["TSDF_DEBUG_INTEGRATE"] = str(v)
            sdf = ts.SDF(args.m, with_color=not args.no_color)
            trk = ts.CameraTracking(sdf=sdf)
            trk.set_K(seq.K)
            sdf.set_timing(True)
            best = None
            for rep in range(args.sweeps):
                sdf.read_timing(reset=True)
                sdf.read_counters(reset=True)
                for k in range(args.frames):
                    trk.set_camera_transformation(seq.R[k], seq.t[k])
                    sdf.set_frame_device(d[k][0].data_ptr(), d[k][1].data_ptr(), d[k][2].data_ptr(), 640, 480)
                    sdf.update(want_stats=False)
                tm, cn = sdf.read_timing(), sdf.read_counters()
                ms = tm["integrate_ms"] / tm["integrate_launches"]
                cms = tm["color_ms"] / tm["color_launches"] if tm.get("color_launches") else 0.0
                if rep > 0 and (best is None or ms < best[0]):
                    best = (ms, cn["n_updated"] / tm["integrate_launches"], cn["integrate_items"] / tm["integrate_launches"], cms)
            print(json.dumps({"scene": scene, "debug": v, "integrate_launch_us": round(best[0] * 1e3, 2),
                              "color_sweep_us": round(best[3] * 1e3, 2),
                              "updated_per_launch": best[1], "items_per_launch": best[2]}), flush=True)
            del trk, sdf


if __name__ == "__main__":
    main()
