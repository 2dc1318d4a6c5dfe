#!/usr/bin/env python3
"""Memory-side counters (vector L1 = TCP, texture addresser = TA, L2 = TCC) of the hot kernels, per launch.

  python3 tools/pmc_memside.py <out_dir> [tag]        (GPU box)

Separate rocprofv3 --pmc passes (never with a trace domain) of tools/bench_kernels.py (fusion-only integrate
launches at the true poses + tracker passes), a handful of counters per pass so that each group fits the block's
slots.  Writes <out_dir>/<tag>_pmc_memside.json: mean per launch and kernel, plus ratios that read directly:
  tcp_busy            TCP_GATE_EN1 / (GRBM_GUI_ACTIVE/8 * 256 CUs)     how long the vector L1s were clocked
  ta_addr_stall       TA_ADDR_STALLED_BY_TC / TA_TA_BUSY               TA waiting for the L1 to take addresses
  ta_data_stall       TA_DATA_STALLED_BY_TC / TA_TA_BUSY
  l2_hit              TCC_HIT / (TCC_HIT + TCC_MISS)
  tcp_read_latency    TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ      cycles, L1 miss -> data back
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CMD = [sys.executable, os.path.join(ROOT, "tools", "bench_kernels.py"), "--frames", "12", "--passes", "8", "--no-track-timing"]
PASS_TIMEOUT = 75
KERNELS = "tsdf::"      # counters are collected for these kernels only (every profiled dispatch is serialised)
PASSES = [
    ["GRBM_GUI_ACTIVE", "TCP_GATE_EN1_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
    ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TOTAL_WAVEFRONTS_sum"],
    ["TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_BUSY_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum"],
    ["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_TAG_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"],
]


QUICK = [4, 5, 6]      # L2 hit rate + memory-side read / write requests by size: the traffic of a launch


def main():
    out = os.path.abspath(sys.argv[1])
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    extra = sys.argv[3:]
    global PASSES
    if extra and extra[0] == "--quick":
        extra = extra[1:]
        PASSES = [PASSES[i] for i in QUICK]
    elif extra and extra[0].startswith("--passes="):
        PASSES = [PASSES[int(i)] for i in extra[0].split("=")[1].split(",")]
        extra = extra[1:]
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    kern = {}
    failed = []
    for i, counters in enumerate(PASSES):
        d = os.path.join(out, "%s_ms_pass%d" % (tag, i))
        shutil.rmtree(d, ignore_errors=True)
        try:      # a counter group the profiler cannot schedule can take forever: give every pass a bounded time
            p = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-include-regex", KERNELS, "--output-format", "csv", "-d", d, "--"] + CMD + extra,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=PASS_TIMEOUT)
        except subprocess.TimeoutExpired:
            failed.append({"counters": counters, "stderr": "timed out after %d s" % PASS_TIMEOUT})
            print("pass", i, counters, "TIMED OUT", flush=True)
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            failed.append({"counters": counters, "stderr": p.stderr[-400:]})
            print("pass", i, counters, "FAILED", flush=True)
            continue
        print("pass", i, "ok", flush=True)
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] not in counters:
                    continue
                name = r["Kernel_Name"].replace("void ", "").split("(")[0]
                if not name.startswith("tsdf::"):
                    continue
                a = kern.setdefault(name, {}).setdefault(r["Counter_Name"], [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        shutil.rmtree(d, ignore_errors=True)
    doc = {"command": "rocprofv3 --pmc <group> --output-format csv -- python3 tools/bench_kernels.py --frames 12 --passes 8 --no-track-timing "
                      + " ".join(extra) + " (one group per pass; tools/pmc_memside.py)",
           "groups": PASSES, "failed_passes": failed, "kernels": {}}
    for name, cs in kern.items():
        m = {c: tot / n for c, (n, tot) in cs.items()}
        m["launches"] = max(n for n, _ in cs.values())
        g = m.get

        def ratio(a, b):
            return g(a) / g(b) if g(a) is not None and g(b) else None
        cu_cycles = g("GRBM_GUI_ACTIVE") / 8.0 * 256.0 if g("GRBM_GUI_ACTIVE") else None
        m["derived"] = {
            "tcp_busy": g("TCP_GATE_EN1_sum") / cu_cycles if cu_cycles and g("TCP_GATE_EN1_sum") is not None else None,
            "ta_addr_stall": ratio("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum"),
            "ta_data_stall": ratio("TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum"),
            "l2_hit": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) else None,
            "tcp_read_latency_cycles": ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
            "hbm_read_bytes": (32.0 * g("TCC_EA0_RDREQ_32B_sum") + 128.0 * g("TCC_EA0_RDREQ_128B_sum")
                               + 64.0 * (g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum") - g("TCC_EA0_RDREQ_128B_sum")))
            if g("TCC_EA0_RDREQ_sum") is not None and g("TCC_EA0_RDREQ_32B_sum") is not None and g("TCC_EA0_RDREQ_128B_sum") is not None else None,
            "hbm_write_bytes": (64.0 * g("TCC_EA0_WRREQ_64B_sum") + 32.0 * (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum")))
            if g("TCC_EA0_WRREQ_sum") is not None and g("TCC_EA0_WRREQ_64B_sum") is not None else None,
            "valu_busy": ratio("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"),
            "waves_waiting": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
            "valu_insts_per_wave": ratio("SQ_INSTS_VALU", "SQ_WAVES"),
        }
        doc["kernels"][name] = m
    with open(os.path.join(out, tag + "_pmc_memside.json"), "w") as f:
        json.dump(doc, f, indent=1)
    for name, m in doc["kernels"].items():
        print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in m["derived"].items()})
    if failed:
        print("FAILED passes:", failed)


if __name__ == "__main__":
    main()
