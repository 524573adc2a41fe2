#!/bin/bash
set -u
out=gpurun_out/r4c9; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@" > $out/$name.log 2>&1; rc=$?; echo "== $name rc=$rc"; tail -n 3 $out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ]; then exit $rc; fi; }
# rehearsal of the multi-rank bench path on one GPU: 2 ranks, gloo rendezvous, both on cuda:0 (smaller batch: two replicas share the card)
run ddp2_bucketed 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 --backend gloo --single-device --no-cpu-baseline
run ddp2_flat 600 env MM_DDP=flat python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 --backend gloo --single-device --no-cpu-baseline
run ddp2_torch 600 env MM_DDP=torch python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 --backend gloo --single-device --no-cpu-baseline
# one rank over RCCL itself (world size 1 group: the communicator, the async all-reduce and work.wait() on the RCCL stream)
run rccl1 600 env WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29520 python3 tools/ddp_overhead.py
ls $out
