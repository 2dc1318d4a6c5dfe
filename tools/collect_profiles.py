#!/usr/bin/env python3
"""Collect the rocprofv3 evidence bench.py's roofline object refers to (run on the GPU box).

  python3 tools/collect_profiles.py <out_dir> [tag]          (then copy <out_dir>/<tag>_* into profiles/)

  1. rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras ...   -> <tag>_bench_kernel_stats.csv (our kernels
     only) and <tag>_bench_under_rocprofv3.json (the bench line of that run)
  2. python3 bench.py (the default command; it measures roofline.traffic itself with two --pmc child passes)
                                                                            -> <tag>_bench.json
  3. tools/pmc_memside.py (separate --pmc passes: SQ, TCP, TCC request sizes) -> <tag>_pmc_memside.json
  4. tools/pmc_calibrate.py (known-byte kernels; needs `make pmc_calibrate`)  -> <tag>_fetch_calibration.json
  6. tools/pmc_ta.py (TA / L1 stall counters of integrate_kernel, one counter per pass) -> <tag>_pmc_ta_integrate.json
  5. rocprofv3 --kernel-trace --stats -- python3 tools/preproc_workload.py  -> <tag>_preproc_kernel_stats.csv
rocprofv3 always gets the program itself after `--`; --pmc is never combined with a trace domain.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = [sys.executable, os.path.join(ROOT, "bench.py")]


def last_json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith('{"metric"')]
    return json.loads(lines[-1]) if lines else None


def main():
    out = os.path.abspath(sys.argv[1])
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    # 1. kernel trace
    d = os.path.join(out, "kt")
    shutil.rmtree(d, ignore_errors=True)
    p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + BENCH +
                       ["--steps", "60", "--warmup", "5", "--no-extras", "--no-cpu-baseline"], cwd="/tmp", env=env,
                       capture_output=True, text=True)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if p.returncode or not stats:
        sys.exit("kernel-trace run failed: " + p.stderr[-2000:])
    with open(stats[0]) as f, open(os.path.join(out, tag + "_bench_kernel_stats.csv"), "w", newline="") as g:
        rd = csv.reader(f)
        wr = csv.writer(g)
        wr.writerow(next(rd))
        for row in rd:
            if "tsdf::" in row[0]:          # the renderer's torch kernels are input generation, not the hot path
                wr.writerow(row)
    line = last_json_line(p.stdout)
    if line:
        with open(os.path.join(out, tag + "_bench_under_rocprofv3.json"), "w") as f:
            json.dump(line, f, indent=1)
    shutil.rmtree(d, ignore_errors=True)
    # 5. the pre-processing stage alone (bilateral grid, then the windowed filter), kernel trace
    with open(os.path.join(out, tag + "_preproc_kernel_stats.csv"), "w", newline="") as g:
        wr = csv.writer(g)
        for mode in ("1", "0"):
            shutil.rmtree(d, ignore_errors=True)
            p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "preproc_workload.py"), mode], cwd="/tmp", env=env, capture_output=True, text=True)
            stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
            if p.returncode or not stats:
                print("preproc kernel trace failed: " + p.stderr[-500:], file=sys.stderr)
                continue
            with open(stats[0]) as f:
                rd = csv.reader(f)
                hdr = next(rd)
                wr.writerow(["grid_filter=" + mode] + hdr)
                for row in rd:
                    if "tsdf::" in row[0] and "fill_kernel" not in row[0]:
                        wr.writerow([""] + row)
    shutil.rmtree(d, ignore_errors=True)
    # 2. the default bench command
    p = subprocess.run(BENCH, cwd=ROOT, env=env, capture_output=True, text=True)
    line = last_json_line(p.stdout)
    if p.returncode or not line:
        sys.exit("bench.py failed: " + p.stderr[-2000:])
    with open(os.path.join(out, tag + "_bench.json"), "w") as f:
        json.dump(line, f, indent=1)
    # 3. / 4.
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_memside.py"), out, tag, "--passes=9,10,0,1,4,5,6,7"],
                   cwd=ROOT, env=env)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_ta.py"), os.path.join(out, tag + "_pmc_ta_integrate.json")], cwd=ROOT, env=env)
    if os.path.exists(os.path.join(ROOT, "build", "pmc_calibrate")):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_calibrate.py"), out], cwd=ROOT, env=env)
    print(json.dumps({"value": line["value"], "traffic": line["roofline"]["traffic"], "frac": line["roofline"]["frac"]}))


if __name__ == "__main__":
    main()
