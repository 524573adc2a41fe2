#!/bin/bash
set -u
out=gpurun_out/r4c6; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 4 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
run tests 1100 python3 -m pytest tests -m gpu -q --durations=5
run bench 600 python3 bench.py
run trace 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_trace -o run -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-alone-pass
python3 tools/trace_last_step.py $out/prof_trace 14 60 --per-queue > $out/step_kernels_S.txt 2>&1
rm -rf $out/prof_trace
ls -la $out
