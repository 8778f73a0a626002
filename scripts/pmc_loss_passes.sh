#!/bin/bash
# SQ counters of the two passes of the fused loss (scripts/bench_logging.py), two PMC passes of 8 SQ slots each.
#   scripts/pmc_loss_passes.sh <tag>   -> gpurun_out/<tag>/pmc_sq{1,2}/...  (summarise with scripts/summarize_pmc.py)
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq1" -o loss -- python3 "$ROOT/scripts/bench_logging.py" --reps 4 > "$OUT/pmc_sq1.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o loss -- python3 "$ROOT/scripts/bench_logging.py" --reps 4 > "$OUT/pmc_sq2.log" 2>&1
echo done > "$OUT/PMC_DONE"
