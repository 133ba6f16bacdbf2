#!/bin/bash
# tools/c4_timeline.sh VARIANT... -- on the GPU box: the C4 step per variant library (tools/variants/lib_V.so) under rocprofv3
# --kernel-trace; prints when each kernel of the LAST step started and ended (us, relative to the step's first kernel)
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the root of the copy of the repository there)}"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf "/tmp/tl4_$v"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl4_$v -o t -- python3 "$GRAFT_REPO_ROOT"/tools/c4_pieces.py "$GRAFT_REPO_ROOT/tools/variants/lib_$v.so" > /dev/null 2>&1 || exit 1
  f=$(find /tmp/tl4_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "plan_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
print(sys.argv[2])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0][-70:]
    print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  vgpr {r.get('VGPR_Count', '?'):>4}  {name}")
PY
done
