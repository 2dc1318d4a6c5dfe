#!/usr/bin/env python3
"""Texture-addresser (TA) and L1 (TCP) stall counters of integrate_kernel, ONE counter per rocprofv3 --pmc pass and only
that kernel profiled (--kernel-include-regex): the four-counter TA group of tools/pmc_memside.py times out on this pool,
single counters on one kernel take seconds.  Workload: tools/bench_kernels.py (fusion-only integrate launches at the
ground-truth poses).  Run on the GPU box:   python3 tools/pmc_ta.py out.json
rocprofv3 gets the program itself after `--`; --pmc is never combined with a trace domain."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum",
            "TA_TOTAL_WAVEFRONTS_sum", "TA_BUFFER_WAVEFRONTS_sum", "TA_BUFFER_READ_WAVEFRONTS_sum", "TA_BUFFER_WRITE_WAVEFRONTS_sum",
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum",
            "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"]
KERNEL = "tsdf::integrate_kernel"
CMD = [sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "--frames", "12", "--passes", "2", "--no-track-timing"]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "pmc_ta.json"
    work = tempfile.mkdtemp(prefix="tsdf_ta_", dir="/tmp")
    res, failed = {}, []
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for c in COUNTERS:
            d = os.path.join(work, c)
            cmd = ["rocprofv3", "--pmc", c, "--kernel-include-regex", KERNEL, "--output-format", "csv", "-d", d, "--"] + CMD
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            except subprocess.TimeoutExpired:
                failed.append({"counter": c, "why": "timed out after 120 s"})
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                failed.append({"counter": c, "why": f"rc {p.returncode}: {p.stderr[-200:]}"})
                continue
            tot, n = 0.0, 0
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == c:
                        tot += float(r["Counter_Value"]); n += 1
            if n:
                res[c] = {"per_launch": tot / n, "launches": n}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    g = lambda k: res.get(k, {}).get("per_launch")
    derived = {}
    if g("TA_TA_BUSY_sum") and g("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, the TA / TCP counters over the 256 CUs
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        derived["kernel_cycles_per_xcd"] = cyc
        derived["ta_busy_fraction_of_the_kernel"] = g("TA_TA_BUSY_sum") / (256.0 * cyc)
        for k, name in (("TA_DATA_STALLED_BY_TC_CYCLES_sum", "ta_waiting_for_data_fraction_of_the_kernel"),
                        ("TCP_PENDING_STALL_CYCLES_sum", "l1_pending_stall_fraction_of_the_kernel"), ("TCP_GATE_EN1_sum", "l1_clocked_fraction_of_the_kernel")):
            if g(k) is not None:
                derived[name] = g(k) / (256.0 * cyc)
        if g("TA_TOTAL_WAVEFRONTS_sum"):
            derived["ta_busy_cycles_per_wavefront_instruction"] = g("TA_TA_BUSY_sum") / g("TA_TOTAL_WAVEFRONTS_sum")
    if g("TA_TA_BUSY_sum"):
        for k, name in (("TA_DATA_STALLED_BY_TC_CYCLES_sum", "ta_waiting_for_data_from_the_L1"), ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "ta_address_path_blocked_by_the_L1")):
            if g(k) is not None:
                derived[name + " / ta_busy"] = g(k) / g("TA_TA_BUSY_sum")
    doc = {"command": "one pass per counter: rocprofv3 --pmc <counter> --kernel-include-regex " + KERNEL + " --output-format csv -- python3 "
                      "tools/bench_kernels.py --frames 12 --passes 2 --no-track-timing   (tools/pmc_ta.py)",
           "kernel": KERNEL + "<true,true,true,true>, 512^3, 640x480, colour on; values per launch, summed over the chip's instances",
           "counters": res, "derived": derived, "failed_passes": failed}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({"derived": derived, "failed": failed}, indent=1))


if __name__ == "__main__":
    main()
