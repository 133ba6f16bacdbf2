#!/bin/bash
# tools/c3_steps_timeline.sh -- on the GPU box: `bench.py --workload c3 --steps 6` under rocprofv3 --kernel-trace; prints start .. end (us)
# and the gaps of the kernels of the last steps
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the root of the copy of the repository there)}"
export TMPDIR=/tmp
D=/tmp/bt_x
rm -rf "$D"
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python3 bench.py --workload c3 --steps 6 --warmup 2 --no-cpu --no-secondary --no-fill > /dev/null 2>&1 || exit 1
python3 - "$(find "$D" -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
ks = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in ks if "pg::" in r["Kernel_Name"]][-10:]
t0 = int(ks[0]["Start_Timestamp"])
prev_end = None
for r in ks:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:6.1f}"
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}  {gap:12}  {r['Kernel_Name'].split('(')[0][-50:]}")
    prev_end = e
PY
