#!/usr/bin/env python3
"""Run build/pmc_calibrate under rocprofv3 --pmc (one counter group per pass) and compare what the counters say with
the bytes each kernel is known to move.  Writes <out>/r02_fetch_calibration.json.

Passes: FETCH_SIZE | WRITE_SIZE | the L2's memory-side read requests by size (TCC_EA0_RDREQ, _32B, _128B) | write
requests (TCC_EA0_WRREQ, _64B).  Derived: read bytes = 32 n32 + 128 n128 + 64 (n - n32 - n128); write bytes =
64 n64 + 32 (n - n64).

  python3 tools/pmc_calibrate.py <out_dir>         (GPU box; `make pmc_calibrate` first)
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "pmc_calibrate")
KERNELS = "cal_"      # counters are collected for these kernels only (every profiled dispatch is serialised)
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
          ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum"],
          ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]]


def collect(cmd_tail, out, tag):
    """Run the passes over `cmd_tail`; returns ({kernel: {counter: [values in launch order]}}, stdout of the last run)."""
    env = dict(os.environ, TMPDIR="/tmp")
    got, stdout = {}, ""
    for i, counters in enumerate(PASSES):
        d = os.path.join(out, "%s_pass%d" % (tag, i))
        shutil.rmtree(d, ignore_errors=True)
        p = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-include-regex", KERNELS, "--output-format", "csv", "-d", d, "--"] + cmd_tail,
                           cwd="/tmp", env=env, capture_output=True, text=True)
        if p.returncode != 0:
            sys.exit("pass %s failed: %s" % (counters, p.stderr[-2000:]))
        stdout = p.stdout
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] not in counters:
                    continue
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                scale = 1024.0 if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") else 1.0     # those two come in KB
                got.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]) * scale)
        shutil.rmtree(d, ignore_errors=True)
    return got, stdout


def derived(v):
    """v: {counter: value} of one launch -> read / write bytes from the request counters."""
    n, n32, n128 = v.get("TCC_EA0_RDREQ_sum", 0.0), v.get("TCC_EA0_RDREQ_32B_sum", 0.0), v.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    w, w64 = v.get("TCC_EA0_WRREQ_sum", 0.0), v.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    return 32.0 * n32 + 128.0 * n128 + 64.0 * (n - n32 - n128), 64.0 * w64 + 32.0 * (w - w64)


def median(x):
    x = sorted(x)
    return x[len(x) // 2] if x else 0.0


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    got, stdout = collect([EXE], out, "cal")
    txt = stdout[stdout.index("{"):]
    known = json.loads(txt[:txt.rindex("}") + 1])
    doc = {"command": "rocprofv3 --pmc <group> --output-format csv -- build/pmc_calibrate, groups: %s (tools/pmc_calibrate.py)" % PASSES,
           "derived": "read bytes = 32 n32 + 128 n128 + 64 (RDREQ - n32 - n128); write bytes = 64 n64 + 32 (WRREQ - n64)",
           "kernels": {}}

    def entry(k_read, k_write, vals):
        rd, wr = derived(vals)
        e = {"known_read": k_read, "known_write": k_write}
        e.update(vals)
        e["derived_read_bytes"], e["derived_write_bytes"] = rd, wr
        if k_read:
            e["FETCH_SIZE_over_known"] = vals.get("FETCH_SIZE", 0.0) / k_read
            e["derived_read_over_known"] = rd / k_read
        if k_write:
            e["WRITE_SIZE_over_known"] = vals.get("WRITE_SIZE", 0.0) / k_write
            e["derived_write_over_known"] = wr / k_write
        return e
    for name, k in known.items():
        vals = got.get(name, {})
        if isinstance(k, dict):
            doc["kernels"][name] = entry(k["read"], k["write"], {c: median(v) for c, v in vals.items()})
        else:     # cal_rmw_items: launched twice per repetition (64 live lanes, then 46) -> split by launch order
            for j, kk in enumerate(k):
                doc["kernels"]["%s[%d live lanes]" % (name, kk["live_lanes"])] = entry(
                    kk["read"], kk["write"], {c: median(v[j::len(k)]) for c, v in vals.items()})
    with open(os.path.join(out, "r02_fetch_calibration.json"), "w") as f:
        json.dump(doc, f, indent=1)
    for name, e in doc["kernels"].items():
        print(name, {k: round(v, 3) for k, v in e.items() if k.endswith("_over_known")})


if __name__ == "__main__":
    main()
