# rocprofv3 kernel-trace stats of the config4 / config5 presets -> gpurun_out/prof_config{4,5}
set -eo pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in config4 config5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$v -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --overlap off --preset $v > $ROOT/gpurun_out/prof_$v.log 2>&1
done
