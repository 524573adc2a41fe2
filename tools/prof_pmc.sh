#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_pmc.sh OUTNAME "COUNTER1 COUNTER2 ..." -- python3 prog args
# Collects PMC counters in their own run (kernel-trace only; never mixed with sys/hip traces).
set -u
name=$1; ctrs=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$name
timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/$name -- "$@" > gpurun_out/$name/run.log 2>&1
echo "prof $name rc=$?"
