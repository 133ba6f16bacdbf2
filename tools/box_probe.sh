#!/bin/bash
# tools/box_probe.sh TAG -- run on the GPU box (through gpurun).  What distinguishes the boxes of the pool on which the
# small-tile emitters (C3, C4) run 10-15 % slower while C2 does not move?  Logs what the box says about itself
# (partition modes, clocks, power cap, firmware) next to the three workloads' times and the concurrent-vs-sequential
# pre-pass A/B, into gpurun_out/box_<TAG>/.
set -u
T=${1:-x}
OUT=gpurun_out/box_$T
mkdir -p $OUT
(rocm-smi --showcomputepartition --showmemorypartition --showclocks --showpower --showmaxpower --showfwinfo --showperflevel 2>&1 | head -120) > $OUT/rocm_smi.txt
(rocminfo 2>&1 | grep -E "Name:|Compute Unit|Max Clock|Cache|Marketing|Uuid|Memory Properties|Size" | head -80) > $OUT/rocminfo.txt
python3 bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
python3 tools/ab_emit.py run_c4 19 6 > $OUT/ab_c4.txt 2>/dev/null
python3 tools/ab_emit.py run_c3 20 8 > $OUT/ab_c3.txt 2>/dev/null
python3 - <<PY
import json
l = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("box $T: c2 %.3f  c3 %.3f (%.3f ms)  c4 %.3f (%.2f ms)" % (l["roofline"]["frac"], l["secondary"]["c3"]["roofline"]["frac"],
      l["secondary"]["c3"]["ms_per_step"], l["secondary"]["c4"]["roofline"]["frac"], l["secondary"]["c4"]["ms_per_step"]))
for f in ("ab_c4", "ab_c3"):
    for line in open("$OUT/%s.txt" % f):
        d = json.loads(line)
        print("  %s %-22s median %.3f ms  min %.3f" % (f, d["variant"], d["median_ms"], d["min_ms"]))
PY
head -40 $OUT/rocm_smi.txt
