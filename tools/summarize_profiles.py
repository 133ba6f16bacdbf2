#!/usr/bin/env python3
"""tools/summarize_profiles.py ROUND -- gpurun_out/prof_<round>/ -> profiles/<round>_* + profiles/pmc_summary.json"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", f"prof_{R}")
DST = os.path.join(ROOT, "profiles")
summary = {}
# the kernel sources the passes were collected for: written on the GPU box by collect_profiles.sh, checked against this tree
SOURCES = open(os.path.join(SRC, "kernel_sources.sha256")).read().split()[0]
from plonk_gadgets_amd import build as pg_build
if SOURCES != pg_build.kernel_sources_sha256():
    print("WARNING: the passes under", SRC, "were collected for other kernel sources than this tree's")


def newest(pattern):
    """gpurun merges runs into the same directory: take the most recent file that matches"""
    return max(glob.glob(pattern), key=os.path.getmtime)


for w in ("c2", "c3", "c4", "c2_values"):
    if not glob.glob(os.path.join(SRC, f"stats_{w}", "*", "*kernel_stats.csv")):
        continue
    st = newest(os.path.join(SRC, f"stats_{w}", "*", "*kernel_stats.csv"))
    shutil.copy(st, os.path.join(DST, f"{R}_{w}_kernel_stats.csv"))
    log = open(os.path.join(SRC, f"stats_{w}.log")).read().strip().splitlines()
    line = [l for l in log if l.startswith("{")][-1]
    open(os.path.join(DST, f"{R}_{w}_bench_under_rocprof.json"), "w").write(line + "\n")
    chunk = json.loads(line)["config"]["items_per_launch"]
    vals = {}
    for kind, cn in (("pmcw", "WRITE_SIZE"), ("pmcf", "FETCH_SIZE")):
        f = newest(os.path.join(SRC, f"{kind}_{w}", "*", "*counter_collection.csv"))
        # every kernel of the one step the PMC pass runs (steps = 1, warmup = 0): the emit launch(es), the inversion
        # pre-pass, the plan and its prefix-sum launch
        keep = [r for r in csv.DictReader(open(f)) if "pg::" in r["Kernel_Name"]]
        with open(os.path.join(DST, f"{R}_{w}_pmc_{cn.lower()}.csv"), "w") as o:
            wr = csv.DictWriter(o, fieldnames=list(keep[0].keys()))
            wr.writeheader()
            wr.writerows(keep)
        vals[cn] = sum(float(r["Counter_Value"]) for r in keep)
    # MI355X_MICROARCH.md: WRITE_SIZE exact (KB) for 16-B-per-lane streaming stores; FETCH_SIZE counts half the bytes
    # of wide streaming reads on gfx950 -> doubled
    summary[w] = {str(chunk): {"kernel": "every kernel of one step", "write_size_kb": vals["WRITE_SIZE"],
                               "fetch_size_kb_raw": vals["FETCH_SIZE"],
                               "hbm_bytes_per_launch": (vals["WRITE_SIZE"] + 2 * vals["FETCH_SIZE"]) * 1024,
                               "round": R, "kernel_sources_sha256": SOURCES,
                               "note": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes; FETCH_SIZE doubled "
                                       "(gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM)"}}
    # launches of one step may overlap (a big-item gadget's pre-pass runs beside its emit launch): the step's duration is
    # the SPAN of its kernels, not the sum of their durations -- from the kernel trace, last step
    tr = newest(os.path.join(SRC, f"stats_{w}", "*", "*kernel_trace.csv"))
    ks = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [r for r in ks if "pg::" in r["Kernel_Name"]]
    # the last step: from the last kernel that begins a step (c2: the pre-pass or the emit launch; c3: the launch that plans,
    # inverts and writes the variable table; c4: the plan kernel)
    first_of = {"c2": ("batch_invert", "emit_kernel"), "c3": ("scalar_mix_vars",), "c4": ("plan_kernel",),
                "c2_values": ("batch_invert", "emit_kernel")}[w]
    firsts = [i for i, r in enumerate(ks) if any(k in r["Kernel_Name"] for k in first_of)]
    start = firsts[-1]
    if w in ("c2", "c2_values") and len(firsts) > 1 and firsts[-2] == start - 1:  # pre-pass and emit launch of the same step
        start = firsts[-2]
    step = ks[start:]
    t0 = int(step[0]["Start_Timestamp"])
    with open(os.path.join(DST, f"{R}_{w}_step_timeline.txt"), "w") as o:
        o.write(f"# {w}: kernels of the last step of `bench.py --workload {w} --steps 5` under rocprofv3 --kernel-trace: start .. end (us)\n")
        for r in step:
            o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}  {r['Kernel_Name'].split('(')[0][-70:]}\n")
        span = (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3
        o.write(f"# span {span:.1f} us; sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3:.1f} us\n")
    print(w, "step span us", span)
    rows = list(csv.DictReader(open(st)))
    for r in rows[:3]:
        print(w, r["Name"][:72], r["Calls"], "avg_us=%.1f" % (float(r["AverageNs"]) / 1e3))
json.dump(summary, open(os.path.join(DST, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: {c: v["hbm_bytes_per_launch"] for c, v in d.items()} for k, d in summary.items()}))
