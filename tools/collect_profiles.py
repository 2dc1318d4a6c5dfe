#!/usr/bin/env python3
"""Collect the rocprofv3 evidence bench.py's roofline object refers to (run on the GPU box).

  python3 tools/collect_profiles.py <out_dir> [tag]

Three separate rocprofv3 runs of the same bench.py command (never --pmc together with a trace domain):
  1. --kernel-trace --stats      -> <tag>_bench_kernel_stats.csv, <tag>_bench.json (the bench line of that run)
  2. --pmc FETCH_SIZE            } -> <tag>_pmc_traffic.json: mean bytes per launch and kernel, and the HBM
  3. --pmc WRITE_SIZE            }    traffic of one integrate launch (clip_rows_kernel + integrate_kernel)
rocprofv3 is started with the program itself after `--` (python3 bench.py ...), as the pool requires.
Copy the three files into profiles/ to have them judged.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = ["python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"]


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT).returncode


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    out = os.path.abspath(sys.argv[1])
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01_final"
    os.makedirs(out, exist_ok=True)
    # 1. kernel trace
    d = os.path.join(out, "kt")
    shutil.rmtree(d, ignore_errors=True)
    log = os.path.join(out, "kt.log")
    rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + BENCH +
             ["--steps", "60", "--warmup", "5"], log)
    stats = glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))
    if rc or not stats:
        sys.exit("kernel-trace run failed, see " + log)
    shutil.copy(stats[0], os.path.join(out, tag + "_bench_kernel_stats.csv"))
    line = [ln for ln in open(log) if ln.startswith('{"metric"')]
    if line:
        with open(os.path.join(out, tag + "_bench.json"), "w") as f:
            json.dump(json.loads(line[-1]), f, indent=1)
    # 2./3. PMC passes
    kernels = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_" + counter)
        shutil.rmtree(d, ignore_errors=True)
        log = os.path.join(out, "pmc_%s.log" % counter)
        rc = run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + BENCH +
                 ["--steps", "20", "--warmup", "2"], log)
        files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
        if rc or not files:
            sys.exit("pmc %s run failed, see %s" % (counter, log))
        acc = {}
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0          # rocprofv3 reports KB
        for k, (n, tot) in acc.items():
            kernels.setdefault(k, {})[counter] = {"launches": n, "mean_bytes": tot / n}

    def tot(kname):
        e = kernels.get(kname, {})
        return sum(e.get(c, {}).get("mean_bytes", 0.0) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    integ = [k for k in kernels if k.startswith("tsdf::integrate_kernel")]
    traffic = tot("tsdf::clip_rows_kernel") + sum(tot(k) for k in integ)
    doc = {
        "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline "
                   "(one counter per pass; tools/collect_profiles.py)",
        "unit": "bytes per launch (rocprofv3 reports KB)",
        "kernels": kernels,
        "integrate_launch_traffic_bytes": traffic,
        "correction": "none applied: the gfx950 1/2-count of FETCH_SIZE holds for wide (16 B/lane) coalesced streaming "
                      "reads; here WRITE_SIZE equals the known write volume (updated voxels x 24 B) and FETCH_SIZE "
                      "exceeds the known read volume (updated voxels x 24 B + 9.8 MB of pixel records), so no halving "
                      "is present for this 8/16-byte RMW + gather pattern",
    }
    with open(os.path.join(out, tag + "_pmc_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({"traffic": traffic, "kernel_stats": os.path.basename(stats[0])}))


if __name__ == "__main__":
    main()
