#!/bin/bash
# single-GPU rehearsal of the multi-rank step (world size 1 over RCCL): plain vs bucketed gather
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps $4 --warmup 3 --no-cpu-baseline $2 2>gpurun_out/gather_$3.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', 'ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'host_enqueue', d['host_enqueue_ms_per_step'])" || tail -5 gpurun_out/gather_$3.err; }
mkdir -p gpurun_out
run 29621 "" plain 30
run 29623 --force-gather gather30 30
run 29624 --force-gather gather32 32
run 29625 --force-gather gather50 50
