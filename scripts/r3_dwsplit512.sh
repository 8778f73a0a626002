#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for ms in 128 64 32 128 64 32 96 48; do
  XFMR_DW_MAXSPLIT=$ms timeout -k 10 200 python bench.py --steps 40 --warmup 10 --spinup-steps 100 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B 512 dw maxsplit $ms', d['ms_per_step'], d['value'], 'resident', d['resident']['ms_per_step'])"
done
