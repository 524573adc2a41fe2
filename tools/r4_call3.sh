#!/bin/bash
# Round-4 GPU call 3: full GPU suite (no -x), rows forward kernel vs the general kernel per stage (training and inference form)
set -u
out=gpurun_out/r4c3; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 6 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
run parity 600 python3 -m pytest tests/test_scan_parity.py -m gpu -q -x
run scan_rows 400 env FWD_VARIANTS=0,16,0x20010,0x10010 python3 tools/bench_scan_bwd.py S 64 0
run scan_rows_B 400 env FWD_VARIANTS=0,16 python3 tools/bench_scan_bwd.py B 32 0
run tests 1100 python3 -m pytest tests -m gpu -q --durations=8 -s
ls -la $out
