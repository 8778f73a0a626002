#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
for args in "--batch 32" "--batch 128" "--batch 512" "--preset config4" "--preset config5"; do
for w in 1 0 1 0; do
  XFMR_ROWSUM_WIDE=$w timeout -k 10 200 python bench.py $args --steps 40 --warmup 10 --spinup-steps 100 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args wide=$w', d['ms_per_step'], d['value'])"
done; done
