cd $GRAFT_REPO_ROOT
for sk in "" fps a2 geo "fps,a2" "fps,a2,geo"; do
  echo "skip=[$sk] $(AMC3D_SKIP=$sk timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/skip.err | tail -1 | cut -c1-40)"
done
