#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for args in "--preset config4" "--preset config5" "--preset config4 --negatives in-batch" "--preset reference-default"; do
  timeout -k 10 200 python bench.py $args --steps 30 --warmup 8 --spinup-steps 60 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$args', d['ms_per_step'], d['value'], [(k['kernel'][22:35], k['avg_launch_ms']) for k in r['kernels']])"
done
