#!/bin/bash
# isolated loss-pass timings at the H = 128 and H = 256 shapes (+ A/B of the H = 256 masked fast path)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "== config 2 shape (H 128, B 512)"; timeout -k 10 200 python scripts/bench_logging.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/logging_h128.log || exit 1
echo "== config 2 shape, BPR head (LSE in the logging pass)"; timeout -k 10 200 python scripts/bench_logging.py --head PairwiseLogisticLoss 2>&1 | grep -v amdgpu.ids | tee gpurun_out/logging_h128_bpr.log || exit 1
echo "== config 5 shape (H 256, 64 x 512, V 1e6, CCL)"; timeout -k 10 300 python scripts/bench_logging.py --hidden 256 --batch 64 --seq-len 512 --items 1000000 --head AlignmentContrastiveLoss --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/logging_h256.log || exit 1
echo "== same, general epilogue (XFMR_LOSS_LOGM256=0)"; XFMR_LOSS_LOGM256=0 timeout -k 10 300 python scripts/bench_logging.py --hidden 256 --batch 64 --seq-len 512 --items 1000000 --head AlignmentContrastiveLoss --reps 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/logging_h256_general.log || exit 1
