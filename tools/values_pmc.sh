#!/bin/bash
# tools/values_pmc.sh [TABLES] -- on the GPU box: tools/values_pmc.py under rocprofv3 --pmc, one pass per counter group (never
# combined with other trace domains; the program directly after `--`), then per TABLE the counters of the dispatches that wrote
# it beside the time those dispatches took IN THE SAME PROCESS: what distinguishes a table that takes 5.6 ms from one that takes
# 6.7?  Output: gpurun_out/values_pmc/summary.json (+ the passes' CSVs).
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the root of the copy of the repository there)}"
T=${1:-6}
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/values_pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
pass() {  # name, counters...
  local name=$1; shift
  rm -rf "/tmp/vp_$name"
  PYTHONPATH="$GRAFT_REPO_ROOT" timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "/tmp/vp_$name" -o p -- \
      python3 "$GRAFT_REPO_ROOT/tools/values_pmc.py" "$T" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr" || { tail -5 "$OUT/$name.stderr"; return 1; }
  cp "$(find "/tmp/vp_$name" -name '*counter_collection.csv' | head -1)" "$OUT/$name.counters.csv"
  cp "$(find "/tmp/vp_$name" -name '*kernel_trace.csv' | head -1)" "$OUT/$name.trace.csv"
  echo "pass $name done"
}
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
pass stall TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum
pass misc TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum GRBM_GUI_ACTIVE
python3 - "$OUT" "$T" <<'PY'
import collections, csv, json, os, sys
out, tables = sys.argv[1], int(sys.argv[2])
summary = {}
for name in ("wr", "stall", "misc"):
    stdout = [l for l in open(os.path.join(out, name + ".stdout")) if l.startswith("{")]
    info = json.loads(stdout[-1])
    calls = info["calls_per_table"]
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(out, name + ".trace.csv")))}
    rows = [r for r in csv.DictReader(open(os.path.join(out, name + ".counters.csv"))) if "EmitMode)3" in r["Kernel_Name"] or "RangeCheckGD, 3" in r["Kernel_Name"]]
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    per_table = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        d = disp.index(int(r["Dispatch_Id"]))
        t = (d // calls) % tables
        per_table[t][r["Counter_Name"]].append(float(r["Counter_Value"]))
        tr = trace.get(r["Dispatch_Id"])
        if tr and r["Counter_Name"] == rows[0]["Counter_Name"]:
            per_table[t]["kernel_us"].append((int(tr["End_Timestamp"]) - int(tr["Start_Timestamp"])) / 1e3)
    summary[name] = {"ms_by_table_events": info["ms_by_table_round1"], "dispatches": len(disp),
                     "per_table": {t: {k: sorted(v)[len(v) // 2] for k, v in c.items()} for t, c in sorted(per_table.items())}}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for name, s in summary.items():
    print("==", name, "ms by table (events, this process):", s["ms_by_table_events"])
    for t, c in s["per_table"].items():
        print("   table", t, {k: (round(v, 1) if k == "kernel_us" else int(v)) for k, v in c.items()})
PY
