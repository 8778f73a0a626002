#!/bin/bash
# packed ragged regime: gradient-pass split plans (XFMR_LOSS_NSPLIT_GRAD) on one box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $ROOT
for i in 1 2; do for g in 0 1 2; do
  if [ $g = 0 ]; then unset XFMR_LOSS_NSPLIT_GRAD; else export XFMR_LOSS_NSPLIT_GRAD=$g; fi
  python bench.py --lengths ml --no-cpu-baseline --no-ragged 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nsplit_grad=$g', d['ms_per_step'], d['value'], [ (k['kernel'][:28], k['avg_launch_ms']) for k in d['roofline']['kernels'][:2]])"
done; done
