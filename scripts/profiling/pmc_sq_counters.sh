# Per-kernel SQ counters (wave cycles, waits, VALU / LDS / MFMA activity) of one bench run: two --pmc passes.
# Run through gpurun from the repo root: gpurun -- bash scripts/profiling/pmc_sq_counters.sh
set -eo pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq1 -o b -- python3 $ROOT/bench.py --steps 3 --warmup 2 --spinup-steps 0 --no-cpu-baseline --overlap off > $ROOT/gpurun_out/pmc_sq1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq2 -o b -- python3 $ROOT/bench.py --steps 3 --warmup 2 --spinup-steps 0 --no-cpu-baseline --overlap off > $ROOT/gpurun_out/pmc_sq2.log 2>&1
