#!/bin/bash
# stream-priority (= HW queue pool) sweep of the pipelined step, with and without an RCCL process group alive
python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range()); s=torch.cuda.Stream(priority=-1); print(s.priority)"
for P in "0,0,0,0" "-1,0,0,0" "0,-1,0,0" "-1,-1,0,0" "0,0,-1,-1" "-1,0,-1,-1"; do
  AMC3D_STREAM_PRIO=$P timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ps_a.log 2> gpurun_out/ps_a.err || exit 1
  A=$(python3 scratch/show_bench.py gpurun_out/ps_a.log | head -1)
  AMC3D_STREAM_PRIO=$P AMC3D_CAPTURE_MODE=thread_local timeout -k 10 200 python scratch/bench_with_pg.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ps_b.log 2> gpurun_out/ps_b.err || exit 1
  B=$(python3 scratch/show_bench.py gpurun_out/ps_b.log | head -1)
  echo "prio $P : plain $A | with PG $B"
done
