#!/bin/bash
# small batches: column splits of the loss passes (one box)
cd "${GRAFT_REPO_ROOT:-.}"
for B in 32 128; do
for ns in 0 2 3 4 6 8 12; do
  if [ $ns = 0 ]; then unset XFMR_LOSS_NSPLIT; else export XFMR_LOSS_NSPLIT=$ns; fi
  timeout -k 10 200 python bench.py --batch $B --steps 40 --warmup 10 --spinup-steps 100 --no-cpu-baseline --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('B $B nsplit $ns', d['ms_per_step'], d['value'], [ (k['kernel'][22:35], k['avg_launch_ms']) for k in r['kernels']])"
done; done
