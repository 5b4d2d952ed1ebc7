"""Module-shaped views of the C-ABI with the reference's pybind11 entry points.

``pointnet2_batch_cuda`` and ``pointops_cuda`` below expose exactly the names and
positional arguments of the reference's two extension modules
(cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24, cpp/pointops/src/pointops_api.cpp:14):
torch tensors in, outputs written in place.  Registering them in ``sys.modules``
under those names lets the reference's own, unmodified Python wrappers
(models/layers/group.py, subsample.py, upsampling.py, cpp/pointops/functions/pointops.py)
run on the MI355X kernels -- see INTEGRATION.md.
"""
import ctypes
import sys
import types

import torch

from . import _lib


def _p(t, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("expected a GPU tensor (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")  # reference: CHECK_CONTIGUOUS -> exit(-1)
    if dtype is not None and t.dtype != dtype:  # reference: data_ptr<float>() / data_ptr<int>() throw on another dtype
        raise RuntimeError(f"expected scalar type {dtype} but found {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def _f(t):
    return _p(t, torch.float32)


def _i(t):
    return _p(t, torch.int32)


def _s(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _call(name, dev_tensor, *args):
    with torch.cuda.device(dev_tensor.device):
        _lib.check(getattr(_lib.load(), name)(*args, _s(dev_tensor)), name)


def _grid_ws(b, n_support, m_queries, device):
    wb = int(_lib.load().amc3d_grid_search_workspace_bytes(b, n_support, m_queries))
    return torch.empty(max(wb, 4), dtype=torch.uint8, device=device), wb


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    work, wb = _grid_ws(b, n, m, xyz.device)
    _call("amc3d_ball_query", xyz, b, n, m, float(radius), nsample, _f(new_xyz), _f(xyz), _i(idx), _p(work), wb)
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _call("amc3d_group_points", points, b, c, n, npoints, nsample, _f(points), _i(idx), _f(out))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _call("amc3d_group_points_grad", grad_out, b, c, n, npoints, nsample, _f(grad_out), _i(idx), _f(grad_points),
          None, 0)
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _call("amc3d_gather_points", points, b, c, n, npoints, _f(points), _i(idx), _f(out))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _call("amc3d_gather_points_grad", grad_out, b, c, n, npoints, _f(grad_out), _i(idx), _f(grad_points))
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    wb = int(_lib.load().amc3d_fps_workspace_bytes(b, n))
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=points.device)
    _call("amc3d_furthest_point_sampling", points, b, n, m, _f(points), _f(temp) if temp is not None else None, _i(idx),
          _p(work), wb)
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    work, wb = _grid_ws(b, m, n, unknown.device)
    _call("amc3d_three_nn", unknown, b, n, m, _f(unknown), _f(known), _f(dist2), _i(idx), _p(work), wb)


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _call("amc3d_three_interpolate", points, b, c, m, n, _f(points), _i(idx), _f(weight), _f(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _call("amc3d_three_interpolate_grad", grad_out, b, c, n, m, _f(grad_out), _i(idx), _f(weight), _f(grad_points),
          None, 0)


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    n, nb = xyz.shape[0], offset.shape[0]
    lib = _lib.load()
    wbytes = int(lib.amc3d_knnquery_workspace_bytes(n, m, nsample, nb))
    work = torch.empty(wbytes, dtype=torch.uint8, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(lib.amc3d_knnquery(m, nsample, n, nb, _f(xyz), _f(new_xyz), _i(offset), _i(new_offset), _i(idx),
                                      _f(dist2), _p(work), wbytes, 0, _s(xyz)), "amc3d_knnquery")


def _module(name, fns):
    mod = types.ModuleType(name)
    mod.__doc__ = "MI355X implementation of the reference's %s extension (amcontrast3d_amd.compat)" % name
    for f in fns:
        setattr(mod, f.__name__, f)
    return mod


pointnet2_batch_cuda = _module("pointnet2_batch_cuda", [
    ball_query_wrapper, group_points_wrapper, group_points_grad_wrapper, gather_points_wrapper,
    gather_points_grad_wrapper, furthest_point_sampling_wrapper, three_nn_wrapper, three_interpolate_wrapper,
    three_interpolate_grad_wrapper])
pointops_cuda = _module("pointops_cuda", [knnquery_cuda])


def register_native_modules():
    """Make ``import pointnet2_batch_cuda`` / ``import pointops_cuda`` resolve to this library."""
    sys.modules["pointnet2_batch_cuda"] = pointnet2_batch_cuda
    sys.modules["pointops_cuda"] = pointops_cuda
