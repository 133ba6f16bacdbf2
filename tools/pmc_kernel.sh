#!/bin/bash
# tools/pmc_kernel.sh WORKLOAD KERNEL_SUBSTRING [OUTDIR] -- on the GPU box: SQ counters of one kernel of `bench.py --workload W`, in
# their own rocprofv3 --pmc passes (which serialise the launches: the kernel is seen alone); prints the per-dispatch averages
W=${1:-c3}; K=${2:-scalar_mix_vars}; OUT=${3:-gpurun_out/pmc_$W}
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-secondary --no-fill"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || exit 1
python3 - "$OUT" "$K" <<'PY'
import collections, csv, glob, json, sys
out, key = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
meta = {}
for f in glob.glob(out + "/pmc*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"vgpr": int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count") or 0), "lds": int(r["LDS_Block_Size"]), "grid": r.get("Grid_Size")}
res = {k: sum(v) / len(v) for k, v in agg.items()}
res.update(meta)
if res.get("SQ_WAVE_CYCLES"):
    wc = res["SQ_WAVE_CYCLES"]
    res["derived"] = {k: round(res[c] / wc, 4) for k, c in (("wait_any", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"), ("issuing_vmem", "SQ_ACTIVE_INST_VMEM"), ("issuing_lds", "SQ_ACTIVE_INST_LDS"), ("lds_stalled", "SQ_WAIT_INST_LDS")) if c in res}
    if res.get("SQ_WAVES"):
        res["derived"]["valu_insts_per_wave"] = round(res["SQ_INSTS_VALU"] / res["SQ_WAVES"], 1)
        res["derived"]["wave_cycles_per_wave"] = round(wc / res["SQ_WAVES"], 1)
print(json.dumps(res, indent=1))
PY
