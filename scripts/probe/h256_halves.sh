#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $ROOT
run() { python bench.py --preset config5 --no-cpu-baseline --no-ragged 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['value'], [(k['kernel'][:24], k['avg_launch_ms']) for k in d['roofline']['kernels'][:2]])"; }
for i in 1 2; do
  unset XFMR_HIP_LIB XFMR_LOSS_NSPLIT_GRAD; run tree
  export XFMR_HIP_LIB=$ROOT/build/libxfmr_hip_h256halves.so
  for g in 1 2 4; do XFMR_LOSS_NSPLIT_GRAD=$g run halves_nsg$g; done
done
