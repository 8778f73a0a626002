#!/bin/bash
# The "other shapes" table of DESIGN.md section 5: one bench line per workload -> gpurun_out/<tag>/shapes/*.json
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG/shapes
mkdir -p "$OUT"
cd "$ROOT"
run() { name=$1; shift; timeout -k 10 240 python bench.py --no-cpu-baseline --no-ragged "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "FAILED $name"; echo "$name done"; }
run b32 --batch 32
run b128 --batch 128
run b256 --batch 256
run b512
run b1024 --batch 1024
run b512_lean --lean
run b128_ml --batch 128 --lengths ml
run config3 --preset config3
run ccl_b128 --batch 128 --loss AlignmentContrastiveLoss
run fp32_b128 --batch 128 --precision fp32
run config4 --preset config4
run config4_inbatch --items 27278 --hidden 256 --layers 6 --inter 1024 --batch 64
run config5 --preset config5
run refdefault --preset reference-default
run config1 --preset config1
run config1_eager --preset config1 --graph off
