#!/bin/bash
# Run GPU steps one after another on the gpurun box; every step gets its own timeout and log under gpurun_out/.
# An ordinary failure (assertion, non-zero exit) does not stop the following steps; a step that times out or is killed
# (exit 124 / 137 / 139) does: nothing else is started on a GPU that may be in a bad state.
#   usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== step $name (timeout ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name rc=$rc"; tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then echo "=== stopping after $name (rc=$rc)"; exit $rc; fi
done
exit 0
