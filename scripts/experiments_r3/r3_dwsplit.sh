#!/bin/bash
# small batches: split count of the weight-gradient GEMMs (one box)
cd "${GRAFT_REPO_ROOT:-.}"
export XFMR_LOSS_NSPLIT=8
for B in 32 128; do
for ms in 128 64 32 16 8; do
  export XFMR_DW_MAXSPLIT=$ms
  timeout -k 10 200 python bench.py --batch $B --steps 40 --warmup 10 --spinup-steps 100 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $B dw maxsplit $ms', d['ms_per_step'], d['value'])"
done; done
