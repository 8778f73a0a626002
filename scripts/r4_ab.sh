#!/bin/bash
# round-4 A/B runs on one box: env switches and one library variant, alternating; one-stream kernel trace of the ragged regime
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4e; mkdir -p $OUT; cd $ROOT
line() { python3 -c "import json,sys; d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', d['ms_per_step'], d['value'], 'resident', d['resident']['ms_per_step'], 'ragged', (d.get('ragged') or {}).get('ms_per_step'), 'reduce', d['roofline'].get('reduction_launch_ms'))"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-ragged > $OUT/base_$i.log 2>/dev/null; line $OUT/base_$i.log base
  XFMR_REDUCE_HALF_EARLY=1 python bench.py --no-cpu-baseline --no-ragged > $OUT/half_$i.log 2>/dev/null; line $OUT/half_$i.log half_early
  XFMR_HIP_LIB=$ROOT/build/libxfmr_hip_dwpf2.so python bench.py --no-cpu-baseline --no-ragged > $OUT/dwpf2_$i.log 2>/dev/null; line $OUT/dwpf2_$i.log dwpf2
done | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for len in ml dense; do
XFMR_DW_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$len -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --no-ragged --overlap off --graph off --lengths $len > $OUT/trace_$len.log 2>&1
done
python3 $ROOT/scripts/prof_top.py $(ls $OUT/trace_ml/*/*kernel_stats.csv $OUT/trace_ml/*kernel_stats.csv 2>/dev/null | head -1) 24 > $OUT/top_ml.txt 2>&1 || true
python3 $ROOT/scripts/prof_top.py $(ls $OUT/trace_dense/*/*kernel_stats.csv $OUT/trace_dense/*kernel_stats.csv 2>/dev/null | head -1) 24 > $OUT/top_dense.txt 2>&1 || true
echo done
