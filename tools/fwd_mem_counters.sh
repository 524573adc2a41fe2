#!/bin/bash
# Memory-side counters of the forward scan at one S stage (default: 14x14), inference form (FWD_CHK=0) and training form (1).
# Few counters per pass: the TCP / TCC blocks take 2-3 at a time ("Request exceeds the capabilities of the hardware to collect").
# usage (GPU box, repo root): tools/fwd_mem_counters.sh OUT [stage]
set -u
out=$1; stage=${2:-2}
export FWD_VARIANTS=0 TMPDIR=/tmp
groups=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
        "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
        "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES")
mkdir -p gpurun_out/$out
for chk in 0 1; do
  export FWD_CHK=$chk
  dirs=""
  for i in "${!groups[@]}"; do
    d=gpurun_out/$out/c${chk}_p$i; mkdir -p $d; dirs="$dirs $d"
    timeout -k 5 100 rocprofv3 --pmc ${groups[$i]} --kernel-trace --output-format csv -d $d -- python3 tools/bench_scan_bwd.py S 64 0 $stage > $d/run.log 2>&1
    rc=$?; echo "chk=$chk pass $i rc=$rc"
    if [ $rc -ne 0 ]; then grep -m3 -i "error\|exceeds" $d/run.log; exit $rc; fi
  done
  python3 tools/pmc_summary.py $dirs > gpurun_out/$out/summary_chk$chk.txt 2>&1
  rm -rf $dirs
done
