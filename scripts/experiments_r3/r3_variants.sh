#!/bin/bash
# isolated loss-pass timings + step A/B for experiment builds: scripts/experiments_r3/r3_variants.sh name1 name2 ...   ("tree" = the product build)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = tree ]; then unset XFMR_HIP_LIB; else export XFMR_HIP_LIB="$PWD/build/libxfmr_hip_$v.so"; fi
  echo "== $v: isolated loss passes"
  timeout -k 10 200 python scripts/bench_logging.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/logging_$v.log || exit 1
done
for r in 1 2; do
for v in "$@"; do
  if [ "$v" = tree ]; then unset XFMR_HIP_LIB; else export XFMR_HIP_LIB="$PWD/build/libxfmr_hip_$v.so"; fi
  timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null > gpurun_out/v_$v.json || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/v_$v.json").read().strip().splitlines()[-1])
print("$v", "h2d", d["ms_per_step"], "resident", d["resident"]["ms_per_step"], "log in-line", d["roofline"]["avg_launch_ms"])
PY
done
done
