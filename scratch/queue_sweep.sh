#!/bin/bash
# hardware-queue plan sweep of the pipelined step: plain and with an RCCL process group alive
for HQ in 1 4; do
for P in "fps,geo" "fps,a2" "fps0,fps1,geo" "fps,a2,b" "pooled"; do
  export GPU_MAX_HW_QUEUES=$HQ
  AMC3D_QUEUES=$P timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ps_a.log 2> gpurun_out/ps_a.err || exit 1
  A=$(python3 scratch/show_bench.py gpurun_out/ps_a.log 2>/dev/null | head -1)
  AMC3D_QUEUES=$P AMC3D_CAPTURE_MODE=thread_local timeout -k 10 200 python scratch/bench_with_pg.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ps_b.log 2> gpurun_out/ps_b.err || exit 1
  B=$(python3 scratch/show_bench.py gpurun_out/ps_b.log 2>/dev/null | head -1)
  echo "hwq $HQ plan $P : plain $A | with PG $B"
done
done
