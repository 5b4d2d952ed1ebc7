set -e
cd $GRAFT_REPO_ROOT
echo S $(timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/chk_S.err | tail -1 | cut -c1-70)
echo S-nojoint $(AMC3D_NO_FPS_JOINT=1 timeout -k 10 300 python bench.py --gpus 1 --steps 32 --warmup 8 --lean 2>gpurun_out/chk_S2.err | tail -1 | cut -c1-70)
echo XLMM $(timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --lean --variant XL --mm --batch 2 --points 64000 2>gpurun_out/chk_XL.err | tail -1 | cut -c1-70)
echo XLMM-nopp $(AMC3D_NO_PINGPONG=1 timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --lean --variant XL --mm --batch 2 --points 64000 2>gpurun_out/chk_XL2.err | tail -1 | cut -c1-70)
echo XLMM-mask $(AMC3D_CU_MASK=geo:0:192 timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --lean --variant XL --mm --batch 2 --points 64000 2>gpurun_out/chk_XL3.err | tail -1 | cut -c1-70)
