#!/bin/bash
# tools/ipl_sweep.sh -- the fused mix's arithmetic launch with other numbers of steps per wave (experiment build with PG_EXP_MIX_IPL_ENV)
export PYTHONPATH=.

for ipl in 16 17 18 19 20 21 24; do
  echo "== ipl $ipl"
  PG_EXP_MIX_IPL=$ipl python tools/mix_phases.py iplenv_stamps 2>&1 | tail -3
done
