#!/bin/bash
# Round-4 GPU call 2: full GPU test suite at HEAD (new oracle-parity, two-stream GEMM, version-bump, no_grad, DDP tests), ubench v2,
# the slow-GEMM overlap evidence, GradSync / DDP host overhead, the extended bench line.
set -u
out=gpurun_out/r4c2; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 6 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
run ubench_fwd_loop 200 tools/ubench/fwd_loop
run tests 1100 python3 -m pytest tests -m gpu -q -x --durations=12 -s
run ddp_overhead 400 python3 tools/ddp_overhead.py
run trace 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_trace -o run -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-alone-pass
python3 tools/trace_kernel_instances.py $out/prof_trace Cijk_Ailk_Bjlk_SB_MT64x64x16_MI16x16x4x1 14 2 > $out/slow_gemm_instances.txt 2>&1
python3 tools/trace_last_step.py $out/prof_trace 14 50 > $out/step_kernels_S.txt 2>&1
rm -rf $out/prof_trace
run bench 600 python3 bench.py
ls -la $out
