#!/bin/bash
# tools/profile_write_path.sh ROUND WORKLOAD -- run on the GPU box (through gpurun) from the repo root.
# The L2's memory-side write path under the emit kernel of a workload (c2 / c4), and -- for the ragged emitter's item
# lookup -- the LDS counters; every --pmc pass on its own (never combined with other trace domains).
# Output under gpurun_out/wp_<round>_<workload>/; tools/summarize_write_path.py -> profiles/<round>_<workload>_write_path_counters.json
set -u
R=${1:-r02}
W=${2:-c4}
export TMPDIR=/tmp
OUT=gpurun_out/wp_${R}_$W
mkdir -p $OUT
ARGS="bench.py --workload $W --log2-batch 18 --steps 1 --warmup 0 --no-cpu --no-secondary --no-fill"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || exit 1
echo "write-path counters of $W in $OUT"
