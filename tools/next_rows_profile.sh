#!/bin/bash
# tools/next_rows_profile.sh [LOG2_BATCH] -- on the GPU box: tools/time_next_rows.py under rocprofv3 --kernel-trace --stats; writes
# gpurun_out/next_rows_timing.json (the script's own lines) and gpurun_out/next_rows_kernel_stats.csv
set -eu
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the root of the copy of the repository there)}"
LG=${1:-18}
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
PYTHONPATH=. python3 tools/time_next_rows.py $LG > "$OUT/next_rows_timing.json" 2> "$OUT/next_rows_timing.err" || exit 1
cd /tmp && rm -rf /tmp/nr_prof
PYTHONPATH="$GRAFT_REPO_ROOT" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nr_prof -o nr -- python3 "$GRAFT_REPO_ROOT/tools/time_next_rows.py" "$LG" > /dev/null 2>&1 || exit 1
f=$(find /tmp/nr_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/next_rows_kernel_stats.csv
cat $OUT/next_rows_timing.json
head -25 $OUT/next_rows_kernel_stats.csv | cut -c1-200
