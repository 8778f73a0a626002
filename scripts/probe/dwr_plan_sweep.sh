#!/bin/bash
# sweep of the ring kernel's slab plan (XFMR_DWR_TOKENS = target tokens per slab, XFMR_DWR_MINWG = least workgroups per weight)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd $ROOT
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['resident']['ms_per_step'], (d.get('ragged') or {}).get('ms_per_step'))"; }
for rep in 1 2; do
for tk in 100000000 4096 3072 2048; do for mw in 32 64; do
  echo "tokens $tk minwg $mw: b512 $(XFMR_DWR_TOKENS=$tk XFMR_DWR_MINWG=$mw one) | b128 $(XFMR_DWR_TOKENS=$tk XFMR_DWR_MINWG=$mw one --batch 128 --no-ragged) | b32 $(XFMR_DWR_TOKENS=$tk XFMR_DWR_MINWG=$mw one --batch 32 --no-ragged)"
done; done
echo "generic plan (tokens 1): b512 $(XFMR_DWR_TOKENS=128 XFMR_DWR_MINWG=100000 one) | b128 $(XFMR_DWR_TOKENS=128 XFMR_DWR_MINWG=100000 one --batch 128 --no-ragged) | b32 $(XFMR_DWR_TOKENS=128 XFMR_DWR_MINWG=100000 one --batch 32 --no-ragged)"
done
