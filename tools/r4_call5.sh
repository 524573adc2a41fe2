#!/bin/bash
set -u
out=gpurun_out/r4c5; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 4 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
run parity 600 python3 -m pytest tests/test_scan_parity.py -m gpu -q -x
run scan_wg 400 env FWD_VARIANTS=0,32,0x60020,0x40020,0x80020,0x4000020,0x4060020 python3 tools/bench_scan_bwd.py S 64 0 0
run scan_wg_B 400 env FWD_VARIANTS=0,32,0x80020,0x40020,0x4000020 python3 tools/bench_scan_bwd.py B 32 0 0
run tests 900 python3 -m pytest tests/test_optim_gpu.py tests/test_full_size_gpu.py -m gpu -q -s
ls -la $out
