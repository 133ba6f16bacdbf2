#!/bin/bash
# tools/profile_tlb.sh ROUND -- run on the GPU box (through gpurun).  Address-translation counters under the emit kernels
# of the three workloads (own --pmc passes): are the small-tile launches (C3, C4) paying for translation misses that the
# big-tile one (C2) does not?  Output under gpurun_out/tlb_<round>/.
set -u
R=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/tlb_$R
mkdir -p $OUT
for w in c2 c3 c4; do
  ARGS="bench.py --workload $w --log2-batch 19 --steps 2 --warmup 1 --no-cpu --no-secondary --no-fill"
  timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $OUT/$w -- python3 $ARGS > $OUT/$w.log 2>&1 || echo "pass failed for $w"
done
python3 - <<PY
import csv, glob, collections
for w in ("c2", "c3", "c4"):
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % w)
    if not fs: print(w, "no output"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=lambda f: __import__("os").path.getmtime(f)))):
        k = r["Kernel_Name"]
        if "emit_kernel" in k or "batch_invert" in k:
            tag = ("invert" if "invert" in k else "emit" + (k.split("GD")[1][:8] if "GD" in k else ""))
            agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(w, k, len(v), sum(v) / len(v))
PY
