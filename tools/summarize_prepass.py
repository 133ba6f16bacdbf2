#!/usr/bin/env python3
"""tools/summarize_prepass.py ROUND [WORKLOAD] -- gpurun_out/prepass_<round>_<workload>/ (tools/profile_prepass.sh)
-> profiles/<round>_<workload>_prepass_counters.json: per kernel of the workload, averages over its dispatches of the SQ
counters (own --pmc passes) and the kernel-trace durations.  When the directory holds two generations of files (the
script was run before and after a kernel change) the OLDEST is reported as "before", the newest as "after"."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
W = sys.argv[2] if len(sys.argv) > 2 else "c3"
SRC = os.path.join(ROOT, "gpurun_out", f"prepass_{R}_{W}")


def kernel_key(name):
    if "batch_invert" in name:
        return "batch_invert_kernel"
    if "emit_kernel" in name:
        for mode, tag in ((", 2>", "emit_kernel<EMIT_ROWS>"), (", 3>", "emit_kernel<EMIT_VARS>")):
            if mode in name:
                return tag
        return "emit_kernel<EMIT_ALL>"
    if "vars_image" in name:
        return "vars_image_kernel"
    if "plan_kernel" in name:
        return "plan_kernel"
    if "scan_" in name:
        return name.split("(")[0].replace("pg::", "")
    return None


def generation(pick):
    out = collections.defaultdict(dict)
    for p in ("pmc1", "pmc2", "pmc3"):
        files = sorted(glob.glob(os.path.join(SRC, p, "*", "*counter_collection.csv")), key=os.path.getmtime)
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[pick])):
            k = kernel_key(r["Kernel_Name"])
            if k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                out[k]["vgpr"] = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0)
                out[k]["lds_bytes"] = int(r["LDS_Block_Size"])
        for (k, c), v in agg.items():
            out[k][c] = sum(v) / len(v)
    files = sorted(glob.glob(os.path.join(SRC, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if files:
        for r in csv.DictReader(open(files[pick])):
            k = kernel_key(r["Name"])
            if k:
                out[k]["avg_us_in_the_concurrent_step"] = float(r["AverageNs"]) / 1e3
    for k, d in out.items():
        if d.get("SQ_WAVES") and d.get("SQ_WAVE_CYCLES"):
            wc = d["SQ_WAVE_CYCLES"]
            d["derived"] = {
                "valu_insts_per_wave": d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"],
                "wave_cycles_issuing_share": d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                "wave_cycles_waiting_share (s_waitcnt / barrier)": d.get("SQ_WAIT_ANY", 0) / wc,
                "wave_cycles_issue_stalled_share": d.get("SQ_WAIT_INST_ANY", 0) / wc,
            }
    return out


n_gen = len(glob.glob(os.path.join(SRC, "pmc1", "*", "*counter_collection.csv")))
res = {"what": f"rocprofv3 --pmc passes (own runs, kernels serialised by the profiler) + one --kernel-trace --stats run over "
               f"python3 bench.py --workload {W} --steps 3 --warmup 1; averages per dispatch; SQ_* cycle counters are quad-cycles",
       "after": generation(-1)}
if n_gen > 1:
    label = ("before (the round-1 kernels, measured at the start of this round with the same script)" if R == "r02" else
             "before (the older generation of files in the directory: the same script run before the last kernel change)")
    res[label] = generation(0)
dst = os.path.join(ROOT, "profiles", f"{R}_{W}_prepass_counters.json")
json.dump(res, open(dst, "w"), indent=1)
for gen, d in res.items():
    if isinstance(d, dict):
        for k, v in d.items():
            if isinstance(v, dict) and "derived" in v:
                print(gen[:6], k, {a: round(b, 3) for a, b in v["derived"].items()}, v.get("avg_us_in_the_concurrent_step"))
