#!/bin/bash
# builds scratch/diag/libamc3d_contrast{1,2}.so (contrast backward without atomics / without gathers) next to the product library
set -e
cd /root/repo/amcontrast3d_amd/csrc
make -s
for d in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function -DAMC_CONTRAST_DIAG=$d -c loss.hip -o /tmp/loss_diag$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/diag/libamc3d_contrast$d.so /tmp/loss_diag$d.o $(ls *.o | grep -v "^loss.o")
done
