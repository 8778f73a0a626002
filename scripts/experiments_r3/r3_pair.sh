#!/bin/bash
# paired weight-gradient launches: tests, then A/B by flag at several batches (one box)
cd "${GRAFT_REPO_ROOT:-.}"
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -4 || exit 1
for args in "--batch 32" "--batch 128" "--batch 256" "--preset config4" "--preset config5"; do
for pair in 1 0 1 0; do
  XFMR_DW_PAIR=$pair timeout -k 10 200 python bench.py $args --steps 40 --warmup 10 --spinup-steps 100 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args pair=$pair', d['ms_per_step'], d['value'])"
done; done
