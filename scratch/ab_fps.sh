# NOTE: the switch this script toggles (an experiment of round 2) was measured and removed again; kept for the record of how it was measured
cd $GRAFT_REPO_ROOT
for w in 0 16 17; do
  export AMC3D_FPS_WAVES=$w
  echo "--- waves=$w"
  timeout -k 10 200 python scratch/fps_bench.py 24000 6000 2>/dev/null | tail -1
  timeout -k 10 200 python scratch/fps_bench.py 6000 1500 2>/dev/null | tail -1
  timeout -k 10 200 python scratch/fps_bench.py 12000 3000 2>/dev/null | tail -1
done
