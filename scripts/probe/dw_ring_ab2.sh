#!/bin/bash
# step-level A/B of ring-kernel builds (build/libxfmr_hip_dwr_*.so) against the generic kernel, alternating on one box
TAG=${1:-r4ring3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
line() { python3 -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', d['ms_per_step'], d['value'], 'resident', d['resident']['ms_per_step'], 'ragged', (d.get('ragged') or {}).get('ms_per_step'))"; }
for i in 1 2 3; do
  XFMR_DW_RING=0 python bench.py --no-cpu-baseline > $OUT/old_$i.log 2>/dev/null; line $OUT/old_$i.log generic
  for v in build/libxfmr_hip_dwr_*.so; do
    n=$(basename $v .so); n=${n#libxfmr_hip_dwr_}
    XFMR_HIP_LIB=$ROOT/$v python bench.py --no-cpu-baseline > $OUT/${n}_$i.log 2>/dev/null; line $OUT/${n}_$i.log $n
  done
done | tee $OUT/ab.txt
