"""Native layer of the reference, rebound onto libamc3d_hip.so.

``pointnet2_cuda`` has the entry points of the reference's pybind11 module
``pointnet2_batch_cuda`` (cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24).
"""
from .pointnet2_batch import pointnet2_cuda
