#!/bin/bash
# attention backward form at small batches (the chip is underfilled: 4 x B workgroups): lock-step vs two roles, one box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $ROOT
for b in 32 64 128 256; do for i in 1 2; do for f in lockstep roles; do
  XFMR_ATTN_BWD_FORM=$f python bench.py --batch $b --no-cpu-baseline --no-ragged --graph off 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch $b $f', d['ms_per_step'], d['value'])"
done; done; done
