#!/bin/bash
# small-batch lines: eager vs hipGraph replay (GraphedStep)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for cfg in "--batch 32" "--batch 128" "--preset reference-default" "--batch 64 --seq-len 50 --hidden 64 --layers 2 --inter 256 --items 1682"; do
  for g in off on; do
    tag=$(echo "$cfg" | tr -d ' -' | cut -c1-24)_$g
    timeout -k 10 300 python bench.py $cfg --graph $g --no-cpu-baseline > gpurun_out/small_$tag.json 2> gpurun_out/small_$tag.err || { tail -20 gpurun_out/small_$tag.err; exit 1; }
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/small_$tag.json").read().strip().splitlines()[-1])
print("$cfg", "graph=$g", "value", d["value"], d["ms_per_step"], "resident", d["resident"]["value"], "|", d["config"].get("launch"))
PY
  done
done
