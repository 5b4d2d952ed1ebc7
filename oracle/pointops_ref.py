"""ORACLE -- test infrastructure only (see oracle/pointops_ref.c).

ctypes front-end of the C restatement.  It exposes the two native module
surfaces of the reference with their exact entry-point names and argument
order, operating on contiguous CPU torch tensors in place:

  * ``pointnet2_batch_cuda`` surface -- openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24
  * ``pointops_cuda`` surface        -- openpoints/cpp/pointops/src/pointops_api.cpp:13-25 (knnquery only)

so that oracle/gen_golden.py can put them under the reference's Python layer,
and oracle/model_ref.py can use them for the CPU restatement of the model.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
import ctypes
import os
import subprocess
import types

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpointops_ref.so")


def build(force=False):
    """Compile oracle/pointops_ref.c with gcc (seconds)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "pointops_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.ref_fps_block_size.restype = ctypes.c_int
    return _lib


def _f(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.device)
    return ctypes.c_void_p(t.data_ptr())


def _i(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu", (t.dtype, t.device)
    return ctypes.c_void_p(t.data_ptr())


_c = ctypes.c_int


# ---- pointnet2_batch_cuda surface ------------------------------------------------
def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    lib().ref_ball_query(_c(b), _c(n), _c(m), ctypes.c_float(radius), _c(nsample), _f(new_xyz), _f(xyz), _i(idx))
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    lib().ref_group_points(_c(b), _c(c), _c(n), _c(npoints), _c(nsample), _f(points), _i(idx), _f(out))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    lib().ref_group_points_grad(_c(b), _c(c), _c(n), _c(npoints), _c(nsample), _f(grad_out), _i(idx), _f(grad_points))
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    lib().ref_gather_points(_c(b), _c(c), _c(n), _c(npoints), _f(points), _i(idx), _f(out))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    lib().ref_gather_points_grad(_c(b), _c(c), _c(n), _c(npoints), _f(grad_out), _i(idx), _f(grad_points))
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    lib().ref_furthest_point_sampling(_c(b), _c(n), _c(m), _f(points), _f(temp), _i(idx))
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    lib().ref_three_nn(_c(b), _c(n), _c(m), _f(unknown), _f(known), _f(dist2), _i(idx))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    lib().ref_three_interpolate(_c(b), _c(c), _c(m), _c(n), _f(points), _i(idx), _f(weight), _f(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    lib().ref_three_interpolate_grad(_c(b), _c(c), _c(n), _c(m), _f(grad_out), _i(idx), _f(weight), _f(grad_points))


# ---- pointops_cuda surface ---------------------------------------------------------
def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    assert nsample <= 100, "reference heap is float[100] (knnquery_cuda_kernel.cu:86-87)"
    lib().ref_knnquery(_c(m), _c(nsample), _f(xyz), _f(new_xyz), _i(offset), _i(new_offset), _i(idx), _f(dist2))


def set_threads(n):
    lib().ref_set_threads(_c(int(n)))
    return int(lib().ref_get_threads())


def fps_block_size(n):
    return int(lib().ref_fps_block_size(_c(n)))


def as_modules():
    """Return (pointnet2_batch_cuda, pointops_cuda) stand-in module objects."""
    m1 = types.ModuleType("pointnet2_batch_cuda")
    for name in ("ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
                 "gather_points_wrapper", "gather_points_grad_wrapper",
                 "furthest_point_sampling_wrapper", "three_nn_wrapper",
                 "three_interpolate_wrapper", "three_interpolate_grad_wrapper"):
        setattr(m1, name, globals()[name])
    m2 = types.ModuleType("pointops_cuda")
    m2.knnquery_cuda = knnquery_cuda
    return m1, m2


# ---- convenience functional forms (allocate outputs like the reference wrappers) ----
def ball_query(radius, nsample, xyz, new_xyz):
    """group.py:177-197 (zero-filled idx, then the kernel)."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    ball_query_wrapper(B, N, M, float(radius), nsample, new_xyz.contiguous(), xyz.contiguous(), idx)
    return idx


def furthest_point_sample(xyz, npoint):
    """subsample.py:78-99 (temp = 1e10)."""
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.int32)
    temp = torch.full((B, N), 1e10, dtype=torch.float32)
    furthest_point_sampling_wrapper(B, N, npoint, xyz.contiguous(), temp, out)
    return out


def three_nn(unknown, known):
    """upsampling.py:14-33 (returns sqrt(dist2), idx)."""
    B, N, _ = unknown.shape
    m = known.shape[1]
    dist2 = torch.empty(B, N, 3, dtype=torch.float32)
    idx = torch.empty(B, N, 3, dtype=torch.int32)
    three_nn_wrapper(B, N, m, unknown.contiguous(), known.contiguous(), dist2, idx)
    return torch.sqrt(dist2), idx


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """cpp/pointops/functions/pointops.py:32-53 (returns idx, sqrt(dist2))."""
    if new_xyz is None:
        new_xyz = xyz
    m = new_xyz.shape[0]
    idx = torch.zeros(m, nsample, dtype=torch.int32)
    dist2 = torch.zeros(m, nsample, dtype=torch.float32)
    knnquery_cuda(m, nsample, xyz.contiguous(), new_xyz.contiguous(), offset.contiguous(), new_offset.contiguous(), idx, dist2)
    return idx, torch.sqrt(dist2)
