#!/bin/bash
# tools/c3_timeline.sh VARIANT... -- on the GPU box: one C3 call per variant library under rocprofv3 --kernel-trace; prints
# when each kernel of the LAST call started and ended (us, relative to the call's first kernel)
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the root of the copy of the repository there)}"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf "/tmp/tl_$v"
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -o t -- python3 "$GRAFT_REPO_ROOT"/tools/c3_pieces.py "$GRAFT_REPO_ROOT/tools/variants/lib_$v.so" ${C3_FORM:-} > /dev/null 2>&1 || exit 1
  f=$(find /tmp/tl_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last arithmetic launch on (the planned call has no plan kernel of its own)
last = max(i for i, r in enumerate(rows) if "scalar_mix_vars_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
print(sys.argv[2])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} us  {name}")
PY
done
