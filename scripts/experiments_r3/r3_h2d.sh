#!/bin/bash
# round 3, first GPU call: sanity + the H2D-inclusive headline at three batches
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
for B in 512 128 32; do
  timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/h2d_b$B.json 2> gpurun_out/h2d_b$B.err || { tail -20 gpurun_out/h2d_b$B.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/h2d_b$B.json").read().strip().splitlines()[-1])
print($B, "value", d["value"], d["ms_per_step"], "resident", d["resident"]["value"], d["resident"]["ms_per_step"], "inline", d["h2d_on_compute_stream"]["value"], "cold", d["cold_start"]["value"])
PY
done
