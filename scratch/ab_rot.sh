set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo split $(timeout -k 10 300 python bench.py --gpus 1 --steps 30 --warmup 8 --lean 2>gpurun_out/rot.err | tail -1 | cut -c1-70)
done
AMC3D_TIMELINE=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 8 --lean 2> gpurun_out/timeline3.err; grep timeline gpurun_out/timeline3.err
echo nograph $(timeout -k 10 300 python bench.py --gpus 1 --steps 10 --warmup 3 --lean --no-graph 2>gpurun_out/rot2.err | tail -1 | cut -c1-70)
echo sync $(AMC3D_FORCE_SYNC_BN=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --lean 2>gpurun_out/rot3.err | tail -1 | cut -c1-70)
