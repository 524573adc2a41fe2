#!/bin/bash
# Round-end measurements on the GPU box (one gpurun call): bench lines of the three single-GPU configs, the rocprofv3 --stats
# summary, the last-step kernel tables, the FETCH / WRITE passes behind profiles/scan_traffic.json, the scan micro-benchmark.
# usage (repo root on the box): tools/final_profiles.sh TAG      -> files under gpurun_out/final_TAG/
set -u
tag=${1:-r2}; out=gpurun_out/final_$tag; mkdir -p $out
export TMPDIR=/tmp
# one MIOpen find-db for the whole set: the first bench process searches (cudnn.benchmark, ~70 s), every later process — profiled
# or not — runs the same solvers (bench.py keeps a database of its own per process otherwise)
export MIOPEN_USER_DB_PATH=/tmp/mm_final_db_$tag; mkdir -p $MIOPEN_USER_DB_PATH
# ... which starts, like bench.py's own private database, from the recorded search (medmamba_amd/tuning/miopen_gfx950)
python3 -c "import sys; sys.path.insert(0, '.'); from medmamba_amd.tuning import seed_miopen_db; print('seeded', seed_miopen_db('$MIOPEN_USER_DB_PATH'))"
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run bench_config3 600 python3 bench.py
tail -1 $out/bench_config3.log > $out/bench_config3.json
run bench_config2_graph 300 python3 bench.py --batch 32 --mode fwd --graph --steps 30 --warmup 5 --no-cpu-baseline
tail -1 $out/bench_config2_graph.log > $out/bench_config2_graph.json
run bench_config2_eager 300 python3 bench.py --batch 32 --mode fwd --steps 30 --warmup 5 --no-cpu-baseline
tail -1 $out/bench_config2_eager.log > $out/bench_config2_eager.json
run bench_config5 400 python3 bench.py --size B --res 384 --batch 32 --steps 10 --warmup 4 --no-cpu-baseline
tail -1 $out/bench_config5.log > $out/bench_config5.json
run scan_S64 200 python3 tools/bench_scan.py S 64 0
run scan_B32 200 python3 tools/bench_scan.py B 32 0
run stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alone-pass
python3 tools/stats_summary.py $out/prof_stats 45 > $out/kernel_stats_incl_warmup.txt
python3 tools/trace_last_step.py $out/prof_stats 14 90 --per-queue > $out/step_kernels_S.txt
python3 tools/trace_kernel_instances.py $out/prof_stats Cijk_Ailk_Bjlk_SB_MT64x64x16_MI16x16x4x1 14 2 > $out/slow_gemm_instances.txt 2>&1
rm -rf $out/prof_stats
run traceB 600 rocprofv3 --kernel-trace --output-format csv -d $out/prof_B -o run -- python3 bench.py --size B --res 384 --batch 32 --steps 4 --warmup 4 --no-cpu-baseline --no-alone-pass
python3 tools/trace_last_step.py $out/prof_B 18 70 > $out/step_kernels_config5.txt
rm -rf $out/prof_B
run pmcF 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmcF -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-alone-pass
run pmcW 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmcW -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-alone-pass
python3 tools/traffic_from_pmc.py $out/pmcF $out/pmcW $out/scan_traffic.json "${MM_HEAD:-unrecorded}" > /dev/null
python3 tools/pmc_summary.py $out/pmcF $out/pmcW > $out/scan_traffic_pmc.txt
rm -rf $out/pmcF $out/pmcW
run scan_layouts 300 env FWD_VARIANTS=0 python3 tools/bench_scan_bwd.py S 64 0
export FWD_VARIANTS=0
tools/prof_pmc.sh final_$tag/sq1 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
export FWD_VARIANTS=0       # (set here, not through `env` behind rocprofv3: the profiler has initialised the GPU by then and an exec is refused)
tools/prof_pmc.sh final_$tag/sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" -- python3 tools/bench_scan_bwd.py S 64 0 0,2
unset FWD_VARIANTS
python3 tools/pmc_summary.py $out/sq1 $out/sq2 > $out/scan_sq_counters.txt 2>&1
rm -rf $out/sq1 $out/sq2
run host_ops_b64 300 python3 tools/host_op_profile.py 64
run host_ops_b8 300 python3 tools/host_op_profile.py 8
run adamw_S 120 python3 tools/bench_adamw.py S
run adamw_B 120 python3 tools/bench_adamw.py B
ls -la $out
