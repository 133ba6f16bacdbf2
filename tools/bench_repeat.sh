#!/bin/bash
# tools/bench_repeat.sh [N] -- on the GPU box: the driver's exact command (`python3 bench.py --gpus 1 --steps 20 --warmup 5`) in N
# fresh processes, one after the other; every line goes to gpurun_out/bench_repeat.jsonl and a min / median / max table of
# every configuration's roofline fraction and ms_per_step to gpurun_out/bench_repeat_summary.json
N=${1:-5}
mkdir -p gpurun_out
: > gpurun_out/bench_repeat.jsonl
for i in $(seq 1 $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 >> gpurun_out/bench_repeat.jsonl 2>> gpurun_out/bench_repeat.err || exit 1
  echo "run $i done"
done
python3 - <<'PY'
import json, statistics
lines = [json.loads(l) for l in open("gpurun_out/bench_repeat.jsonl") if l.startswith("{")]
def col(get):
    v = sorted(get(l) for l in lines)
    return {"min": v[0], "median": statistics.median(v), "max": v[-1], "all": v}
out = {"runs": len(lines), "command": "python3 bench.py --gpus 1 --steps 20 --warmup 5 (fresh process each)",
       "c2": {"frac": col(lambda l: l["roofline"]["frac"]), "ms_per_step": col(lambda l: l["ms_per_step"])}}
for k in ("c3", "c4", "c2_values"):
    out[k] = {"frac": col(lambda l: l["secondary"][k]["roofline"]["frac"]), "ms_per_step": col(lambda l: l["secondary"][k]["ms_per_step"])}
nr = [l["secondary"]["next_rows"] for l in lines if "materialize" in l["secondary"].get("next_rows", {})]
if nr:
    out["materialize_ms"] = sorted(x["materialize"]["ms"]["median"] for x in nr)
    out["permutation_ms"] = sorted(x["permutation"]["ms"]["median"] for x in nr)
json.dump(out, open("gpurun_out/bench_repeat_summary.json", "w"), indent=1)
print(json.dumps({k: (v["frac"] if isinstance(v, dict) and "frac" in v else v) for k, v in out.items()}, indent=1))
PY
