#!/bin/bash
# Per-kernel stall counters over the kernels of a training step run ALONE (single stream): tools/step_counters.sh OUT
set -u
out=$1; export TMPDIR=/tmp MM_TWO_STREAMS=0
groups=("SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
mkdir -p gpurun_out/$out; dirs=""
for i in "${!groups[@]}"; do
  d=gpurun_out/$out/p$i; mkdir -p $d; dirs="$dirs $d"
  timeout -k 5 240 rocprofv3 --pmc ${groups[$i]} --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alone-pass > $d/run.log 2>&1
  rc=$?; echo "pass $i rc=$rc"; if [ $rc -ne 0 ]; then grep -m3 -i "error\|exceeds" $d/run.log; exit $rc; fi
done
python3 tools/pmc_by_kernel.py $dirs > gpurun_out/$out/by_kernel.txt 2>&1
rm -rf $dirs
