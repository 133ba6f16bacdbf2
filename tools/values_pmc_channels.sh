#!/bin/bash
# tools/values_pmc_channels.sh [TABLES] -- as tools/values_pmc.sh, but the raw per-instance counters of the L2 channels (TCC_EA0_WRREQ
# and its DRAM credit stalls without the _sum reduction): are a slow table's stalls spread over all channels or piled on a few?
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the root of the copy of the repository there)}"
T=${1:-6}
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/values_pmc_ch"
rm -rf "$OUT" /tmp/vpc; mkdir -p "$OUT"
cd /tmp
PYTHONPATH="$GRAFT_REPO_ROOT" timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL --kernel-trace --output-format csv -d /tmp/vpc -o p -- \
    python3 "$GRAFT_REPO_ROOT/tools/values_pmc.py" "$T" > "$OUT/stdout" 2> "$OUT/stderr" || { tail -5 "$OUT/stderr"; exit 1; }
f=$(find /tmp/vpc -name '*counter_collection.csv' | head -1)
head -3 "$f" | cut -c1-400
python3 - "$f" "$OUT" "$T" <<'PY'
import collections, csv, json, sys
f, out, tables = sys.argv[1], sys.argv[2], int(sys.argv[3])
info = json.loads([l for l in open(out + "/stdout") if l.startswith("{")][-1])
calls = info["calls_per_table"]
rows = [r for r in csv.DictReader(open(f)) if "RangeCheckGD, 3" in r["Kernel_Name"] or "EmitMode)3" in r["Kernel_Name"]]
print("columns:", list(rows[0].keys()))
disp = sorted({int(r["Dispatch_Id"]) for r in rows})
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    d = disp.index(int(r["Dispatch_Id"]))
    if d // calls < tables:  # first round only
        continue
    per[(d // calls) % tables][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ms by table:", info["ms_by_table_round1"])
res = {}
for t, c in sorted(per.items()):
    line = {}
    for k, v in c.items():
        v = sorted(v)
        line[k] = {"instances": len(v), "min": v[0], "median": v[len(v) // 2], "max": v[-1], "sum": sum(v)}
    res[t] = line
    print("table", t, {k: (x["instances"], int(x["min"]), int(x["median"]), int(x["max"])) for k, x in line.items()})
json.dump({"ms_by_table": info["ms_by_table_round1"], "per_table": res}, open(out + "/summary.json", "w"), indent=1)
PY
