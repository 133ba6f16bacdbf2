#!/usr/bin/env python3
"""tools/summarize_write_path.py ROUND WORKLOAD -- gpurun_out/wp_<round>_<workload>/ -> profiles/<round>_<workload>_write_path_counters.json
(sums over the dispatches of the workload's dominant emit kernel; derived shares as in profiles/r01_c2_write_path_counters.json)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, W = sys.argv[1], sys.argv[2]
SRC = os.path.join(ROOT, "gpurun_out", f"wp_{R}_{W}")
KEY = {"c2": "RangeCheckGD", "c4": "MaxBoundGD", "c3": "ScalarMixGD"}[W]
out = {"what": f"rocprofv3 --pmc passes (own runs) over python3 bench.py --workload {W} --log2-batch 18 --steps 1 --warmup 0; sums over "
               f"the pg::emit_kernel<{KEY}...> dispatches", "box_note": "one box of the pool; boxes differ (DESIGN.md section 4)"}
for p in ("p1", "p2", "p3"):
    f = max(glob.glob(os.path.join(SRC, p, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if "emit_kernel" in r["Kernel_Name"] and KEY in r["Kernel_Name"]:
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
st = max(glob.glob(os.path.join(SRC, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)
for r in csv.DictReader(open(st)):
    if "emit_kernel" in r["Name"] and KEY in r["Name"]:
        out["kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
        break
line = [l for l in open(os.path.join(SRC, "trace.log")).read().splitlines() if l.startswith("{")][-1]
rf = json.loads(line)["roofline"]
out["algorithmic_bytes_per_launch"] = rf["algorithmic_bytes_per_launch"]
g = out.get
d = {}
if g("TCC_EA0_WRREQ_sum"):
    d["bytes_from_wrreq"] = g("TCC_EA0_WRREQ_64B_sum", 0) * 64 + (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum", 0)) * 32
    d["share_of_64B_requests"] = g("TCC_EA0_WRREQ_64B_sum", 0) / g("TCC_EA0_WRREQ_sum")
    d["wrreq_per_tcc_cycle"] = g("TCC_EA0_WRREQ_sum") / g("TCC_CYCLE_sum") if g("TCC_CYCLE_sum") else None
    d["wrreq_stall_share_of_tcc_cycles"] = g("TCC_EA0_WRREQ_STALL_sum", 0) / g("TCC_CYCLE_sum") if g("TCC_CYCLE_sum") else None
if g("TCC_BUSY_sum") and g("TCC_CYCLE_sum"):
    d["tcc_busy_share"] = g("TCC_BUSY_sum") / g("TCC_CYCLE_sum")
    d["dram_credit_stall_share"] = g("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 0) / g("TCC_CYCLE_sum")
if g("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
if g("SQ_WAVE_CYCLES"):
    d["wave_cycles_waiting_share"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
    d["wave_cycles_issuing_share"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
if "kernel_avg_ms" in out:
    d["gbps_from_kernel_trace"] = out["algorithmic_bytes_per_launch"] / out["kernel_avg_ms"] / 1e6
out["derived"] = d
dst = os.path.join(ROOT, "profiles", f"{R}_{W}_write_path_counters.json")
json.dump(out, open(dst, "w"), indent=1)
print(dst, json.dumps(d))
