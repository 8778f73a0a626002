#!/bin/bash
# Collect the judged measurement artefacts of one round on the GPU box (run through gpurun):
#   scripts/collect_profiles.sh <tag>            e.g. r01
# writes under gpurun_out/<tag>/: bench.json (the bench line incl. cpu_baseline), kernel-trace stats of the same
# bench command, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over scripts/bench_logging.py for the HBM
# traffic of the dominant kernel. Copy the summaries into profiles/ with scripts/summarize_profiles.py.
set -eo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 420 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
# per-kernel durations: ONE stream (no side-stream logging pass, weight-gradient GEMMs in line), so that no kernel's
# duration contains another's; the overlapped run the bench line is measured in follows
XFMR_DW_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
  python3 "$ROOT/bench.py" --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off > "$OUT/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_overlap" -o bench -- \
  python3 "$ROOT/bench.py" --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap on > "$OUT/trace_overlap.log" 2>&1
# HBM traffic of the two loss passes in isolation (scripts/bench_logging.py): separate --pmc passes, as the guide prescribes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o loss -- \
    python3 "$ROOT/scripts/bench_logging.py" --reps 4 > "$OUT/pmc_$c.log" 2>&1
done
# the other BASELINE configs (one stream each): configs 4 and 5, and config 2 at the reference's default batch 32
for v in config4 config5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$v" -o bench -- \
    python3 "$ROOT/bench.py" --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off --graph off --preset $v > "$OUT/trace_$v.log" 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b32" -o bench -- \
  python3 "$ROOT/bench.py" --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off --graph off --batch 32 > "$OUT/trace_b32.log" 2>&1
# SQ counters of the same kernels (instruction mix, wait / stall shares, matrix-core busy cycles): two more passes
"$ROOT/scripts/pmc_loss_passes.sh" "$TAG"
# HBM traffic of EVERY kernel of the step (FETCH_SIZE / WRITE_SIZE / L2 hits, separate passes): the dominant kernel's `traffic`
"$ROOT/scripts/pmc_traffic.sh" "$TAG"
echo done > "$OUT/DONE"
