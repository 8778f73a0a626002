set -eo pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r01
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_MFMA" -o loss -- python3 "$ROOT/scripts/bench_loss.py" --reps 4 > "$OUT/pmc_MFMA.log" 2>&1
