#!/bin/bash
# SQ counters of EVERY kernel of the training step (one short bench run under rocprofv3 --pmc).
#   scripts/pmc_step.sh <tag>  -> gpurun_out/<tag>/pmc_step/...  (summarise: scripts/summarize_pmc_step.py <tag>)
set -eo pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export XFMR_DW_SIDE=0  # one stream: counters per kernel, not per overlap
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --kernel-trace --output-format csv -d "$OUT/pmc_step" -o step -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off > "$OUT/pmc_step.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/pmc_step2" -o step -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off > "$OUT/pmc_step2.log" 2>&1
echo done > "$OUT/PMC_STEP_DONE"
