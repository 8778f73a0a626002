#!/bin/bash
# A/B of the LDS-DMA ring weight-gradient kernel (csrc/dw_ring.hip) against the generic split-K one, alternating on one box,
# and a one-stream kernel trace of each: scripts/probe/dw_ring_ab.sh [tag]
TAG=${1:-r4ring}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
line() { python3 -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', d['ms_per_step'], d['value'], 'resident', d['resident']['ms_per_step'], 'ragged', (d.get('ragged') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  XFMR_DW_RING=0 python bench.py --no-cpu-baseline > $OUT/old_$i.log 2>/dev/null; line $OUT/old_$i.log generic
  [ -f build/libxfmr_hip_dwr64.so ] && { XFMR_HIP_LIB=$ROOT/build/libxfmr_hip_dwr64.so python bench.py --no-cpu-baseline > $OUT/r64_$i.log 2>/dev/null; line $OUT/r64_$i.log ring64; }
  python bench.py --no-cpu-baseline > $OUT/ring_$i.log 2>/dev/null; line $OUT/ring_$i.log ring
done | tee $OUT/ab.txt
for b in 32 128; do
  XFMR_DW_RING=0 python bench.py --no-cpu-baseline --no-ragged --batch $b > $OUT/old_b$b.log 2>/dev/null; line $OUT/old_b$b.log generic_b$b
  python bench.py --no-cpu-baseline --no-ragged --batch $b > $OUT/ring_b$b.log 2>/dev/null; line $OUT/ring_b$b.log ring_b$b
done | tee -a $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
XFMR_DW_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off --graph off > $OUT/trace.log 2>&1
python3 $ROOT/scripts/prof_top.py $(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1) 14 | tee $OUT/top.txt
