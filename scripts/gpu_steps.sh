#!/bin/bash
# Run GPU steps one after another; stop at the first step that timed out or was killed (never start another GPU step
# after one that hung). A step that merely FAILS (exit 1) does not stop the sequence.
# Usage: scripts/gpu_steps.sh "cmd1" "cmd2" ...
mkdir -p gpurun_out
rc_all=0
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "=== step $i: $cmd" | tee -a gpurun_out/steps.log
  bash -o pipefail -c "$cmd"
  rc=$?
  echo "=== step $i exit $rc" | tee -a gpurun_out/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then
    echo "step $i timed out / was killed: stopping" | tee -a gpurun_out/steps.log
    exit $rc
  fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
