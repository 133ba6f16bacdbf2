#!/bin/bash
# tools/profile_prepass.sh ROUND [WORKLOAD] -- run on the GPU box (through gpurun) from the repo root.
# Counter evidence for the inversion pre-pass (pg::batch_invert_kernel) of a workload (default c3): occupancy and
# instruction mix in their own rocprofv3 --pmc passes (never combined with other trace domains), plus a kernel trace
# with timestamps that shows how the pre-pass, the plan kernel and the emit kernel overlap.  Output under
# gpurun_out/prepass_<round>_<workload>/; tools/summarize_prepass.py turns it into profiles/<round>_<workload>_prepass_counters.json.
set -u
R=${1:-r02}
W=${2:-c3}
export TMPDIR=/tmp
OUT=gpurun_out/prepass_${R}_$W
mkdir -p $OUT
ARGS="bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-secondary --no-fill"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1 || exit 1
echo "profiled the pre-pass of $W into $OUT"
