#!/bin/bash
# The driver's multi-GPU launch of bench.py, for N ranks of one node (RCCL over xGMI):
#   bash scripts/scale_run.sh 8 [bench.py flags...]
# With no multi-GPU node: the SAME flow with N processes on ONE GPU -- torch.distributed over gloo as the side channel, the library's own
# communicator (fwgpu_dist_*) over the shared-memory stand-in of tests/fake_rccl (host-synchronous: a functional run, not a rate):
#   FAKE=1 bash scripts/scale_run.sh 2 --steps 4 --warmup 1 --batch 8192 --no-cpu-baseline
N=${1:-2}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
PORT=${MASTER_PORT:-29577}
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ -n "$FAKE" ]; then
  export FWGPU_RCCL_LIBRARY=$R/tests/fake_rccl/libfwgpu_fakerccl.so
  exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT $R/bench.py --gpus $N \
       --dist-backend gloo --same-device --library-comm "$@"
fi
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT $R/bench.py --gpus $N "$@"
