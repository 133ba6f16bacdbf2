#!/bin/bash
# tools/collect_profiles.sh ROUND -- run on the GPU box (through gpurun) from the repo root.
# rocprofv3 kernel-trace/stats of bench.py for every workload, then WRITE_SIZE and FETCH_SIZE in their own passes
# (PMC passes never combined with other trace domains), into gpurun_out/prof_<round>/.  tools/summarize_profiles.py
# turns them into profiles/<round>_*.csv + profiles/pmc_summary.json.
set -u
R=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$R
mkdir -p $OUT
python3 -c "from plonk_gadgets_amd import build; print(build.kernel_sources_sha256())" > $OUT/kernel_sources.sha256
for w in c2 c3 c4 c2_values; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu --no-secondary --no-fill > $OUT/stats_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcw_$w -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu --no-secondary --no-fill > $OUT/pmcw_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcf_$w -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu --no-secondary --no-fill > $OUT/pmcf_$w.log 2>&1
  echo "profiled $w"
done
