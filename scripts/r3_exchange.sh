#!/bin/bash
# One rank on RCCL with the collectives of the N > 1 step forced on: packed (one all-gather) against
# the two-collective form and against no exchange, at the shard sizes of configs[2] over 8 / 2 GPUs.
set -e
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--gpus 1 --steps 200 --warmup 5 --cpu-rows 0 --f64-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --profile-steps 0"
for rows in 125 500; do
  python3 bench.py $COMMON --rows $rows > gpurun_out/x_none_$rows.json
  for mode in packed two; do
    if [ $mode = packed ]; then export MPSFR_BENCH_PACKED_EXCHANGE=1; else unset MPSFR_BENCH_PACKED_EXCHANGE; fi
    MPSFR_BENCH_FORCE_EXCHANGE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
      --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) bench.py $COMMON --rows $rows \
      > gpurun_out/x_${mode}_$rows.json
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/x_*.json')):
    ln = [l for l in open(f) if l.startswith('{')]
    o = json.loads(ln[-1]); print(f, round(o['value'] / 1e6, 3), o['ms_per_step'])
PY
