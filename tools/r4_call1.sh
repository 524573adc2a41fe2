#!/bin/bash
# Round-4 GPU call 1: ubench of the forward inner loop forms, parity of the re-laid B/C tile, A/B of the forward kernel against the
# round-3 build, SQ counters, kernel trace of the step for the slow-GEMM hunt, baseline bench line.
set -u
out=gpurun_out/r4c1; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 4 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
run ubench_fwd_loop 120 tools/ubench/fwd_loop
run parity 600 python3 -m pytest tests/test_scan_parity.py -m gpu -x -q
run scan_new 300 env FWD_VARIANTS=0 python3 tools/bench_scan_bwd.py S 64 0
run scan_base 300 env FWD_VARIANTS=0 MM_HIP_LIB=medmamba_amd/lib/libmedmamba_hip_r3base.so python3 tools/bench_scan_bwd.py S 64 0
run scan_new2 300 env FWD_VARIANTS=0 python3 tools/bench_scan_bwd.py S 64 0 0,2
tools/prof_pmc.sh r4c1/sq1 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
tools/prof_pmc.sh r4c1/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
python3 tools/pmc_summary.py $out/sq1 $out/sq2 > $out/scan_sq_counters.txt 2>&1
rm -rf $out/sq1 $out/sq2
run trace 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_trace -o run -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-alone-pass
python3 tools/trace_last_step.py $out/prof_trace 14 60 > $out/step_kernels_S.txt 2>&1
python3 tools/trace_kernel_instances.py $out/prof_trace Cijk_Ailk_Bjlk_SB_MT64x64x16 14 4 > $out/slow_gemm_instances.txt 2>&1
python3 tools/trace_kernel_instances.py $out/prof_trace Cijk_ 14 1 > $out/all_gemm_instances.txt 2>&1
rm -rf $out/prof_trace
run bench 600 python3 bench.py
ls -la $out
