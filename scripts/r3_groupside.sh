#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for args in "--batch 512" "--batch 1024"; do
for gs in 1 0 1 0 1 0; do
  XFMR_DW_GROUP_SIDE=$gs timeout -k 10 200 python bench.py $args --steps 40 --warmup 10 --spinup-steps 100 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args group_side=$gs', d['ms_per_step'], d['value'], 'resident', d['resident']['ms_per_step'])"
done; done
