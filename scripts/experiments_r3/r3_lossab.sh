#!/bin/bash
# isolated loss-pass timings for experiment builds at the H = 128 and config-5 shapes: scripts/experiments_r3/r3_lossab.sh tree name ...
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONUNBUFFERED=1
for v in "$@"; do
  if [ "$v" = tree ]; then unset XFMR_HIP_LIB; else export XFMR_HIP_LIB="$PWD/build/libxfmr_hip_$v.so"; fi
  echo "== $v: config 2 shape"; timeout -k 10 200 python scripts/bench_logging.py --reps 12 2>&1 | grep -v "amdgpu.ids\|library" || exit 1
  echo "== $v: config 2 shape, BPR"; timeout -k 10 200 python scripts/bench_logging.py --reps 12 --head PairwiseLogisticLoss 2>&1 | grep -v "amdgpu.ids\|library" || exit 1
  echo "== $v: config 5 shape, CCL"; timeout -k 10 300 python scripts/bench_logging.py --hidden 256 --batch 64 --seq-len 512 --items 1000000 --head AlignmentContrastiveLoss --reps 8 2>&1 | grep -v "amdgpu.ids\|library" || exit 1
  echo "== $v: config 4 shape (in-batch, InfoNCE masked)"; timeout -k 10 300 python scripts/bench_logging.py --hidden 256 --batch 64 --seq-len 200 --items 27278 --reps 8 2>&1 | grep -v "amdgpu.ids\|library" || exit 1
done
