#!/bin/bash
set -u
out=gpurun_out/r4c10; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 3 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
run glue_tests 600 python3 -m pytest tests/test_glue_gpu.py tests/test_modules_gpu.py -m gpu -q -x
run glue_new 300 python3 tools/bench_glue.py
run glue_prev 300 env MM_HIP_LIB=medmamba_amd/lib/libmedmamba_hip_prev.so python3 tools/bench_glue.py
run bench 400 python3 bench.py --no-cpu-baseline
ls -la $out
