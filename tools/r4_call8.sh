#!/bin/bash
set -u
out=gpurun_out/r4c8; mkdir -p $out
export TMPDIR=/tmp
export FWD_VARIANTS=0
tools/prof_pmc.sh r4c8/sq1 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
tools/prof_pmc.sh r4c8/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
tools/prof_pmc.sh r4c8/sq3 "SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_SMEM" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
python3 tools/pmc_summary.py $out/sq1 $out/sq2 $out/sq3 > $out/scan_sq_counters.txt 2>&1
rm -rf $out/sq1 $out/sq2 $out/sq3
ls $out
