#!/bin/bash
# rehearsal of the driver's N = 2 launch of bench.py on the ONE GPU of a test box: both ranks on device 0, gloo transport
# (ARCO_DIST_BACKEND / ARCO_FORCE_DEVICE, arco_amd/dist.py) - everything but the transport is the N > 1 code path of the bench
o=gpurun_out; mkdir -p $o
ARCO_DIST_BACKEND=gloo ARCO_FORCE_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > $o/r06_bench_n2_rehearsal.json 2> $o/r06_bench_n2_rehearsal.err
echo "rc=$?"; tail -c 1500 $o/r06_bench_n2_rehearsal.json; tail -5 $o/r06_bench_n2_rehearsal.err
