set -eo pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in drop nodrop; do
  extra=""; [ $v = nodrop ] && extra="--no-dropout"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ab_$v -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --overlap off $extra > $ROOT/gpurun_out/ab_$v.log 2>&1
done
