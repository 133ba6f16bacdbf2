#!/bin/bash
# tools/bench_timeline.sh WORKLOAD [ENV=VALUE...] -- on the GPU box: `bench.py --workload W --steps 5` under rocprofv3
# --kernel-trace; prints start .. end (us) of every kernel of the last step
W=${1:-c3}; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
D=/tmp/bt_$$
rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python3 bench.py --workload $W --steps 5 --warmup 1 --no-cpu --no-secondary --no-fill > /dev/null 2>&1 || exit 1
python3 - "$(find $D -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
ks = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in ks if "pg::" in r["Kernel_Name"]]
first = [i for i, r in enumerate(ks) if "plan_kernel" in r["Kernel_Name"]]
first = first[-1] if first else [i for i, r in enumerate(ks) if "emit_kernel" in r["Kernel_Name"]][-1]
t0 = int(ks[first]["Start_Timestamp"])
for r in ks[first:]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} .. {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}  {r['Kernel_Name'].split('(')[0][-60:]}")
PY
