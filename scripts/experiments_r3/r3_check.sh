#!/bin/bash
# GPU tests + smoke + the bench line at two batches (gpurun_out/)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; rc=$?
tail -5 gpurun_out/gputests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
for B in ${BATCHES:-512 128}; do
  timeout -k 10 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.json 2> gpurun_out/bench_b$B.err || { tail -20 gpurun_out/bench_b$B.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b$B.json").read().strip().splitlines()[-1])
r=d["roofline"]
print($B, "value", d["value"], d["ms_per_step"], "resident", d["resident"]["value"], d["resident"]["ms_per_step"], "inline", d["h2d_on_compute_stream"]["value"], "cold", d["cold_start"]["value"], "| dominant", r["kernel"][:40], r["avg_launch_ms"], r["frac"], "overl", r.get("avg_launch_ms_overlapped"))
PY
done
