#!/bin/bash
# one-stream kernel traces of the small shapes (gpurun_out/trace_small/<name>/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/trace_small; cd /tmp && export TMPDIR=/tmp
for v in "b32:--batch 32" "b128:--batch 128" "config4:--preset config4"; do
  name=${v%%:*}; args=${v#*:}
  rm -rf $ROOT/gpurun_out/trace_small/$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/trace_small/$name -o bench -- \
    python3 $ROOT/bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline --overlap off --graph off $args > $ROOT/gpurun_out/trace_small/$name.log 2>&1 || exit 1
done
