for cfg in "2 0 0.7" "2 0 0.6" "2 0 0.5" "2 0 0.8" "4 0 0.7" "4 0 0.8" "4 0 0.6" "8 1 1.1"; do
  set -- $cfg
  echo "== SUB=$1 EXTRA=$2 SCALE=$3"
  AMC3D_KG_SUB=$1 AMC3D_KG_EXTRA=$2 AMC3D_KG_SCALE=$3 timeout -k 10 100 python scratch/knn_bench.py 2>&1 | grep -v amdgpu.ids
done
