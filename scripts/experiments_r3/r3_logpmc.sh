#!/bin/bash
# SQ counters of the logging pass in the harness and in the bench step (one box), to see what differs
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-.}
OUT=$ROOT/gpurun_out/logpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
timeout -k 10 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/harness -o h -- python3 $ROOT/scripts/bench_logging.py --reps 4 > $OUT/harness.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/bench -o b -- python3 $ROOT/bench.py --steps 5 --warmup 2 --spinup-steps 0 --no-cpu-baseline --overlap off --graph off > $OUT/bench.log 2>&1 || exit 1
cd $ROOT
python - <<'PY'
import csv,glob,collections
for ctx in ('harness','bench'):
    f=glob.glob(f'gpurun_out/logpmc/{ctx}/**/*counter_collection.csv',recursive=True)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'loss_main_dma_kernel<128, -3>' in k or 'loss_main_dma_kernel<128, 7>' in k:
            acc[k[:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(ctx,k,{c:round(sum(x[-3:])/len(x[-3:])/1e6,2) for c,x in v.items()}, 'n',len(next(iter(v.values()))))
PY
