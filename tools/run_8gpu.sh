#!/bin/bash
# The N-GPU run of bench.py exactly as the driver launches it (one rank per GPU over RCCL / xGMI), with everything the first real
# run could trip over pinned down: loopback rendezvous, dmabuf IPC, a bounded wait, asynchronous RCCL errors surfaced.
#     tools/run_8gpu.sh [N = 8] [extra bench.py arguments ...]        e.g.  tools/run_8gpu.sh 4 --steps 10 --warmup 2
# Prints ONE JSON line on stdout (the bench contract); rank 0 also writes one line per regime - strong kodak24, sharded clic41,
# throughput - to stderr.  bench.py itself sets the device from LOCAL_RANK BEFORE init_process_group and asserts that the
# communicator has N ranks on N distinct GPUs (`gpus_active`).
set -euo pipefail
N="${1:-8}"; shift || true
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0          # the host driver only supports dmabuf IPC (RCCL's P2P buffers)
export TORCH_NCCL_ASYNC_ERROR_HANDLING=1     # a failed collective aborts the process instead of hanging the others
export TORCH_NCCL_BLOCKING_WAIT=0
export NCCL_DEBUG="${NCCL_DEBUG:-WARN}"
export GPU_MAX_HW_QUEUES="${GPU_MAX_HW_QUEUES:-8}"   # more hardware queues: the library's side streams alias less (ccd_concurrent_streams)
PORT="${MASTER_PORT:-$((20000 + RANDOM % 20000))}"
have=$(python -c 'import torch; print(torch.cuda.device_count())')
if [ "$have" -lt "$N" ]; then echo "run_8gpu.sh: $N ranks asked for, $have GPUs visible" >&2; exit 2; fi
exec timeout "${BENCH_TIMEOUT:-1500}" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" \
    --master-addr 127.0.0.1 --master-port "$PORT" bench.py --gpus "$N" --steps "${STEPS:-20}" --warmup "${WARMUP:-3}" "$@"
