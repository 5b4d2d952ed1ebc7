#!/bin/bash
# which pipeline part costs the overlapped step its time?  (parts left out: timing only, results stale)
for S in none fps a2 geo "fps,a2" "a2,geo" "fps,a2,geo"; do
  AMC3D_SKIP=$S timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/sk.log 2> gpurun_out/sk.err || { tail -3 gpurun_out/sk.err; exit 1; }
  echo "skip $S : $(python3 scratch/show_bench.py gpurun_out/sk.log 2>/dev/null | head -1)"
done
