"""Point operators on MI355X: torch.autograd front-ends over the C-ABI (include/amc3d.h).

These mirror, name for name, the reference's Python wrappers of its CUDA
extensions (argument meaning, dtypes, shapes, contiguity asserts, outputs):

    ball_query, grouping_operation, gather_operation   openpoints/models/layers/group.py:76-203
    furthest_point_sample                               openpoints/models/layers/subsample.py:76-106
    three_nn, three_interpolate, three_interpolation    openpoints/models/layers/upsampling.py:11-102
    knnquery                                            openpoints/cpp/pointops/functions/pointops.py:32-56

PyTorch is only plumbing here (device memory, the current HIP stream, autograd
bookkeeping); every operator runs a hand-written gfx950 kernel.  CPU tensors are
rejected: there is no fallback path.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib, timing


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


_dedicated = []  # (handle, ExternalStream): kept alive for the life of the process


def _destroy_dedicated():
    if not _dedicated:
        return
    try:
        torch.cuda.synchronize()
        for handle, _ in _dedicated:
            _lib.load().amc3d_stream_destroy(handle)
    finally:
        _dedicated.clear()


def dedicated_stream(device=None, first_cu=0, n_cus=0):
    """A torch stream with a hardware queue of its own (amc3d_stream_create_dedicated): for the FPS launches of a
    pipelined loop, which otherwise stall whichever stream shares their queue for milliseconds.  n_cus > 0 restricts
    the stream's kernels to the CUs [first_cu, first_cu + n_cus) of the mask."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(_lib.load().amc3d_stream_create_masked(ctypes.byref(handle), int(first_cu), int(n_cus)),
                   "stream_create_masked")
    s = torch.cuda.ExternalStream(handle.value, device=dev)
    if not _dedicated:
        import atexit
        atexit.register(_destroy_dedicated)
    _dedicated.append((handle, s))
    return s


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("amcontrast3d_amd operators run on the GPU only (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


def mixed_precision():
    """True inside torch.autocast('cuda', dtype=torch.bfloat16): the pointwise convolutions then run in bf16 compute /
    fp32 accumulate (csrc/gemm_bf16.hip) while every tensor stays fp32 in memory (main_AA.py:389-394 use_amp)"""
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


def _pw_forward(lib, bf16, B, Cin, Cout, P, x, w2, bias, y, what="pointwise_conv_forward"):
    """y = w2 . x (+ bias) through the fp32 or bf16-compute kernels; the fp32 short deep layers get scratch for their
    split-K partial products (amc3d_pointwise_conv_forward_ws)"""
    bptr = _ptr(bias) if bias is not None else None
    if not bf16:
        wsf = int(lib.amc3d_pointwise_conv_forward_workspace_bytes(B, Cin, Cout, P, int(bias is not None)))
        if wsf:
            work = torch.empty(wsf, dtype=torch.uint8, device=x.device)
            _lib.check(lib.amc3d_pointwise_conv_forward_ws(B, Cin, Cout, P, _ptr(x), _ptr(w2), bptr, _ptr(y), _ptr(work), wsf,
                                                           _stream(x)), what)
            return
    _lib.check(_pw(lib, bf16)[0](B, Cin, Cout, P, _ptr(x), _ptr(w2), bptr, _ptr(y), _stream(x)), what)


def _pw(lib, bf16):
    """(forward, workspace_bytes, backward) entry points of the pointwise conv in the requested arithmetic"""
    if bf16:
        return (lib.amc3d_pointwise_conv_forward_bf16, lib.amc3d_pointwise_conv_workspace_bytes_bf16,
                lib.amc3d_pointwise_conv_backward_bf16)
    return lib.amc3d_pointwise_conv_forward, lib.amc3d_pointwise_conv_workspace_bytes, lib.amc3d_pointwise_conv_backward


def _need_dtype(dtype, **named):
    """The C-ABI takes raw pointers: a float64 / bf16 coordinate or an int64 index would be reinterpreted silently.
    The reference's pybind layer throws on data_ptr<float>() / data_ptr<int>() of another dtype (ball_query.cpp:29-38);
    so does this."""
    for name, t in named.items():
        if t is not None and t.dtype != dtype:
            raise RuntimeError(f"expected {name} to be {dtype}, got {t.dtype}")


# tests only: {sequence number of the max-pool in forward order: arg-max (B,C,M) uint8} while a dict is installed
# (pool_log(d)); the oracle then routes its max-pool gradients through the same elements (tests/test_gpu_fullsize.py)
_pool_log = None
_pool_seq = 0


def pool_log(d):
    """install (dict) or remove (None) the arg-max log; resets the forward-order counter"""
    global _pool_log, _pool_seq
    _pool_log, _pool_seq = d, 0


def _next_pool_seq():
    global _pool_seq
    _pool_seq += 1
    return _pool_seq - 1


def _grid_ws(b, n_support, m_queries, device):
    """caller-owned scratch of the grid searches (ball query, 3-NN); the library falls back to the all-pairs
    kernels by itself where a grid does not pay (small clouds)"""
    wb = int(_lib.load().amc3d_grid_search_workspace_bytes(b, n_support, m_queries))
    return torch.empty(max(wb, 4), dtype=torch.uint8, device=device), wb


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3) support, new_xyz (B,npoint,3) centres -> idx (B,npoint,nsample) int32"""
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        _need_gpu(xyz, new_xyz)
        _need_dtype(torch.float32, xyz=xyz, new_xyz=new_xyz)
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.empty(B, npoint, nsample, dtype=torch.int32, device=xyz.device)
        work, wb = _grid_ws(B, N, npoint, xyz.device)
        with torch.cuda.device(xyz.device), timing.span("ball_query", (B * N + B * npoint) * 12 + idx.numel() * 4):
            _lib.check(_lib.load().amc3d_ball_query(B, N, npoint, float(radius), int(nsample), _ptr(new_xyz),
                                                    _ptr(xyz), _ptr(idx), _ptr(work), wb, _stream(xyz)), "ball_query")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)"""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        _need_gpu(features, idx)
        _need_dtype(torch.float32, features=features)
        _need_dtype(torch.int32, idx=idx)
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty(B, C, nfeatures, nsample, dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device), timing.span("group_points", features.numel() * 4 + idx.numel() * 4 + output.numel() * 4):
            _lib.check(_lib.load().amc3d_group_points(B, C, N, nfeatures, nsample, _ptr(features), _ptr(idx),
                                                      _ptr(output), _stream(features)), "group_points")
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.detach().contiguous()
        with torch.cuda.device(grad_out.device), timing.span("group_points_grad", grad_out_data.numel() * 4 + idx.numel() * 4 + grad_features.numel() * 4):
            work = torch.empty(B * C * N, dtype=torch.float32, device=grad_out.device)
            _lib.check(_lib.load().amc3d_group_points_grad(B, C, N, npoint, nsample, _ptr(grad_out_data), _ptr(idx),
                                                           _ptr(grad_features), _ptr(work), work.numel() * 4,
                                                           _stream(grad_out)), "group_points_grad")
        return grad_features, None


grouping_operation = GroupingOperation.apply


class GatherOperation(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)"""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        _need_gpu(features, idx)
        _need_dtype(torch.float32, features=features)
        _need_dtype(torch.int32, idx=idx)
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty(B, C, npoint, dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device), timing.span("gather_points", features.numel() * 4 + idx.numel() * 4 + output.numel() * 4):
            _lib.check(_lib.load().amc3d_gather_points(B, C, N, npoint, _ptr(features), _ptr(idx), _ptr(output),
                                                       _stream(features)), "gather_points")
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.detach().contiguous()
        with torch.cuda.device(grad_out.device), timing.span("gather_points_grad", grad_out_data.numel() * 4 + idx.numel() * 4 + grad_features.numel() * 4):
            _lib.check(_lib.load().amc3d_gather_points_grad(B, C, N, npoint, _ptr(grad_out_data), _ptr(idx),
                                                            _ptr(grad_features), _stream(grad_out)), "gather_points_grad")
        return grad_features, None


gather_operation = GatherOperation.apply


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32, first index 0"""
        assert xyz.is_contiguous()
        _need_gpu(xyz)
        _need_dtype(torch.float32, xyz=xyz)
        B, N, _ = xyz.size()
        output = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        # the reference's (B,N) scratch of running minima (filled with 1e10) lives in registers (N <= 24576) or in
        # the workspace (N <= 262144); only larger clouds need the buffer itself
        temp = None
        if N > 262144:
            temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        lib = _lib.load()
        wb = int(lib.amc3d_fps_workspace_bytes(B, N))
        with torch.cuda.device(xyz.device), timing.span("furthest_point_sampling", B * N * 12 + B * int(npoint) * 4):
            work = torch.empty(max(wb, 4), dtype=torch.uint8, device=xyz.device)
            _lib.check(_lib.load().amc3d_furthest_point_sampling(B, N, int(npoint), _ptr(xyz),
                                                                 _ptr(temp) if temp is not None else None,
                                                                 _ptr(output), _ptr(work), wb,
                                                                 _stream(xyz)), "furthest_point_sampling")
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,N,3), known (B,M,3) -> dist (B,N,3) (euclidean, not squared), idx (B,N,3) int32"""
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        _need_gpu(unknown, known)
        _need_dtype(torch.float32, unknown=unknown, known=known)
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)
        work, wb = _grid_ws(B, m, N, unknown.device)
        with torch.cuda.device(unknown.device), timing.span("three_nn", (B * N + B * m) * 12 + B * N * 24):
            _lib.check(_lib.load().amc3d_three_nn(B, N, m, _ptr(unknown), _ptr(known), _ptr(dist2), _ptr(idx),
                                                  _ptr(work), wb, _stream(unknown)), "three_nn")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, features, idx, weight):
        """features (B,C,M), idx (B,n,3) int32, weight (B,n,3) -> (B,C,n)"""
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        _need_gpu(features, idx, weight)
        _need_dtype(torch.float32, features=features, weight=weight)
        _need_dtype(torch.int32, idx=idx)
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device), timing.span("three_interpolate", features.numel() * 4 + idx.numel() * 8 + output.numel() * 4):
            _lib.check(_lib.load().amc3d_three_interpolate(B, c, m, n, _ptr(features), _ptr(idx), _ptr(weight),
                                                           _ptr(output), _stream(features)), "three_interpolate")
        return output

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.detach().contiguous()
        with torch.cuda.device(grad_out.device), timing.span("three_interpolate_grad", grad_out_data.numel() * 4 + idx.numel() * 8 + grad_features.numel() * 4):
            work = torch.empty(B * c * m, dtype=torch.float32, device=grad_out.device)
            _lib.check(_lib.load().amc3d_three_interpolate_grad(B, c, n, m, _ptr(grad_out_data), _ptr(idx),
                                                                _ptr(weight), _ptr(grad_features), _ptr(work),
                                                                work.numel() * 4, _stream(grad_out)),
                       "three_interpolate_grad")
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class ThreeInterpolateAdd(Function):
    """base (B,C,n) + three_interpolate(features (B,C,m), idx, weight): the interpolated branch of a FeaturePropogation
    conv added onto the skip branch in the interpolation kernel itself (no (B,C1+C2,n) concatenation is ever built)"""

    @staticmethod
    def forward(ctx, features, idx, weight, base):
        _need_gpu(features, idx, weight, base)
        _need_dtype(torch.float32, features=features, weight=weight, base=base)
        _need_dtype(torch.int32, idx=idx)
        features, idx, weight, base = features.contiguous(), idx.contiguous(), weight.contiguous(), base.contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        assert base.shape == (B, c, n)
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        output = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        with torch.cuda.device(features.device), timing.span("three_interpolate", features.numel() * 4 + idx.numel() * 8 + output.numel() * 8):
            _lib.check(_lib.load().amc3d_three_interpolate_add(B, c, m, n, _ptr(features), _ptr(idx), _ptr(weight), _ptr(base),
                                                               _ptr(output), _stream(features)), "three_interpolate_add")
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        m = ctx.m
        B, c, n = grad_out.size()
        grad_out_data = grad_out.detach().contiguous()
        grad_features = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device), timing.span("three_interpolate_grad", grad_out_data.numel() * 4 + idx.numel() * 8 + grad_features.numel() * 4):
            work = torch.empty(B * c * m, dtype=torch.float32, device=grad_out.device)
            _lib.check(_lib.load().amc3d_three_interpolate_grad(B, c, n, m, _ptr(grad_out_data), _ptr(idx), _ptr(weight),
                                                                _ptr(grad_features), _ptr(work), work.numel() * 4,
                                                                _stream(grad_out)), "three_interpolate_grad")
        return grad_features, None, None, grad_out


three_interpolate_add = ThreeInterpolateAdd.apply


def three_interpolation(unknown_xyz, known_xyz, know_feat):
    """Inverse-distance 3-NN interpolation (upsampling.py:92-102)."""
    dist, idx = three_nn(unknown_xyz, known_xyz)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    return three_interpolate(know_feat, idx, weight)


_grid_cache = None  # {(support ptr, n, offset ptr, segments, stream): workspace} while knn_grid_reuse() is active


class knn_grid_reuse:
    """with knn_grid_reuse(): k-NN searches over the same support tensor (same storage, same offsets) reuse the
    cell grid the first of them built -- the caller guarantees the support is not modified inside the block."""

    def __enter__(self):
        global _grid_cache
        self.prev = _grid_cache
        if _grid_cache is None:  # re-entrant: an enclosing block's grids stay visible
            _grid_cache = {}
        return self

    def __exit__(self, *exc):
        global _grid_cache
        _grid_cache = self.prev
        return False


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz, offset, new_offset):
        """xyz (n,3), new_xyz (m,3), offset/new_offset (b) int32 cumulative ends
        -> idx (m,nsample) int32, dist (m,nsample) (euclidean)"""
        if new_xyz is None:
            new_xyz = xyz
        assert xyz.is_contiguous() and new_xyz.is_contiguous()
        _need_gpu(xyz, new_xyz, offset, new_offset)
        _need_dtype(torch.float32, xyz=xyz, new_xyz=new_xyz)
        _need_dtype(torch.int32, offset=offset, new_offset=new_offset)
        nsample = int(nsample)
        n, m, nb = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
        idx = torch.empty(m, nsample, dtype=torch.int32, device=xyz.device)
        dist2 = torch.empty(m, nsample, dtype=torch.float32, device=xyz.device)
        lib = _lib.load()
        wbytes = int(lib.amc3d_knnquery_workspace_bytes(n, m, nsample, nb))
        offset, new_offset = offset.contiguous(), new_offset.contiguous()
        # inside `with knn_grid_reuse():` searches over the same support set share one cell grid
        key = (xyz.data_ptr(), n, offset.data_ptr(), nb, _stream(xyz).value)
        cached = _grid_cache.get(key) if _grid_cache is not None else None
        gridded = bool(lib.amc3d_knnquery_uses_grid(m, nsample, n, nb))  # else: all-pairs kernel, no grid in `work`
        reuse = gridded and cached is not None and cached.numel() >= wbytes
        work = cached if reuse else torch.empty(wbytes, dtype=torch.uint8, device=xyz.device)
        if _grid_cache is not None and gridded and not reuse:
            _grid_cache[key] = work
        with torch.cuda.device(xyz.device), timing.span("knnquery", (n + m) * 12 + m * nsample * 8):
            _lib.check(lib.amc3d_knnquery(m, nsample, n, nb, _ptr(xyz), _ptr(new_xyz), _ptr(offset),
                                          _ptr(new_offset), _ptr(idx), _ptr(dist2), _ptr(work), work.numel(),
                                          int(reuse), _stream(xyz)), "knnquery")
        ctx.mark_non_differentiable(idx)
        if _knn_squared:  # knnquery_squared(): the plan code wants no root (one elementwise launch per search on the geometry queue)
            return idx, dist2
        return idx, torch.sqrt(dist2)

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None


knnquery = KNNQuery.apply
_knn_squared = False


def knnquery_squared(nsample, xyz, new_xyz, offset, new_offset):
    """knnquery that leaves the distances SQUARED, as the kernel (and the reference's knnquery_cuda) produces them: the
    reference's wrapper takes the root in a separate elementwise op (pointops.py:55), which the loss's plan never looks at"""
    global _knn_squared
    prev, _knn_squared = _knn_squared, True
    try:
        return KNNQuery.apply(nsample, xyz, new_xyz, offset, new_offset)
    finally:
        _knn_squared = prev


# ----------------------------------------------------------------------------------------------
# adaptive-margin contrastive loss (no native counterpart in the reference: it runs these as
# torch ops + a Python loop -- AMContrast3D/MarginContrast.py, AEF/ambiguity.py, AEF/utils.py)
# ----------------------------------------------------------------------------------------------
def _nbr_view(neighbor_idx):
    """(pointer to first used column, k, row stride) of an int32 index that may be a column slice
    idx[:, 1:] of a contiguous (m, K) tensor -- no copy (the reference calls .contiguous())."""
    assert neighbor_idx.dtype == torch.int32 and neighbor_idx.dim() == 2
    if neighbor_idx.stride(1) != 1:
        neighbor_idx = neighbor_idx.contiguous()
    return _ptr(neighbor_idx), neighbor_idx.shape[1], neighbor_idx.stride(0), neighbor_idx


def vote_labels(labels0, neighbor_idx, num_classes):
    """labels0 (n0) int32 classes of the full-resolution points, neighbor_idx (m,kr) int32 -> (m) int32:
    arg-max of the mean one-hot label over the kr neighbours (AEF/utils.py:29-41)."""
    _need_gpu(labels0, neighbor_idx)
    assert labels0.dtype == torch.int32 and neighbor_idx.dtype == torch.int32 and neighbor_idx.is_contiguous()
    m, kr = neighbor_idx.shape
    out = torch.empty(m, dtype=torch.int32, device=labels0.device)
    with torch.cuda.device(labels0.device), timing.span("vote_labels", m * kr * 8 + m * 4):
        _lib.check(_lib.load().amc3d_vote_labels(m, kr, int(num_classes), _ptr(labels0), _ptr(neighbor_idx), _ptr(out),
                                                 _stream(labels0)), "vote_labels")
    return out


def posmask_from_labels(labels, neighbor_idx):
    """labels (m) int32, neighbor_idx (m,k) int32 (may be idx[:, 1:]) -> (m,k) bool"""
    _need_gpu(labels, neighbor_idx)
    nptr, k, stride, keep = _nbr_view(neighbor_idx)
    m = labels.shape[0]
    out = torch.empty(m, k, dtype=torch.bool, device=labels.device)
    with torch.cuda.device(labels.device), timing.span("posmask", m * k * 9 + m * 4):
        _lib.check(_lib.load().amc3d_posmask(m, k, stride, _ptr(labels), nptr, _ptr(out), _stream(labels)), "posmask")
    return out


_CCTYPE = {"Method1": 1, "Method2": 2, "Method3": 3}


def ambiguity(p, posmask, neighbor_idx, cctype, beta):
    """p (m,3), posmask (m,k) bool, neighbor_idx (m,k) int32 -> a (m) fp32 (AEF/ambiguity.py:11-71)"""
    _need_gpu(p, posmask, neighbor_idx)
    assert p.is_contiguous() and posmask.is_contiguous() and posmask.dtype == torch.bool
    nptr, k, stride, keep = _nbr_view(neighbor_idx)
    m = p.shape[0]
    lib = _lib.load()
    wbytes = int(lib.amc3d_ambiguity_workspace_bytes(m))
    work = torch.empty(wbytes, dtype=torch.uint8, device=p.device)
    a = torch.empty(m, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device), timing.span("ambiguity", m * (12 + 4) + m * k * (4 + 1), moved=m * (12 + 4) + m * k * (4 + 1 + 12)):
        _lib.check(lib.amc3d_ambiguity(m, k, stride, _CCTYPE[cctype], float(beta), _ptr(p), _ptr(posmask), nptr,
                                       _ptr(a), _ptr(work), wbytes, _stream(p)), "ambiguity")
    return a


def select_anchors(a):
    """a (m) fp32 -> int32 list of the anchors with 0 < a <= 1 (MarginContrast.py:250-252), ascending:
    [0] = count, [1..count] = ids (the tail of the tensor is scratch).  Part of a stage's plan: it depends on
    coordinates and labels only."""
    _need_gpu(a)
    _need_dtype(torch.float32, a=a)
    assert a.is_contiguous() and a.dim() == 1
    m = a.shape[0]
    lib = _lib.load()
    n = int(lib.amc3d_select_anchors_ints(m))
    sel = torch.empty(n, dtype=torch.int32, device=a.device)
    with torch.cuda.device(a.device), timing.span("select_anchors", m * 12):
        _lib.check(lib.amc3d_select_anchors(m, _ptr(a), _ptr(sel), n, _stream(a)), "select_anchors")
    return sel


def contrast_csr(neighbor_idx, anchors):
    """Reverse lists of a loss stage's k-NN edges: neighbor_idx (m,k) int32 (may be idx[:, 1:]), anchors =
    select_anchors(a) -> int32 [rev_start (m+1) | rev_edge (m*k)]: for every row n the positions i*k + j, ascending, of the
    selected anchors i whose j-th neighbour is n.  Part of a stage's plan; ContrastStage's backward gathers along it
    instead of scattering with float atomics."""
    _need_gpu(neighbor_idx, anchors)
    _need_dtype(torch.int32, anchors=anchors)
    nptr, k, stride, keep = _nbr_view(neighbor_idx)
    m = neighbor_idx.shape[0]
    lib = _lib.load()
    rev = torch.empty(m + 1 + m * k, dtype=torch.int32, device=anchors.device)
    wbytes = int(lib.amc3d_contrast_csr_workspace_bytes(m))
    work = torch.empty(wbytes, dtype=torch.uint8, device=anchors.device)
    with torch.cuda.device(anchors.device), timing.span("contrast_csr", m * k * 12):
        _lib.check(lib.amc3d_contrast_csr(m, k, stride, nptr, _ptr(anchors), _ptr(rev), _ptr(work), wbytes,
                                          _stream(anchors)), "contrast_csr")
    return rev


def contrast_mutual(neighbor_idx, a, dist2=None):
    """The mutual-edge structure of a loss stage's k-NN graph (csrc/csr.hip amc3d_contrast_mutual): neighbor_idx (m,k) int32
    (may be idx[:, 1:]), a (m) the stage's ambiguities -> (mutual (m,k) uint8, rev int32 [rev_start (m+1) | rev_edge (m*k)]).
    mutual[i,s] = 1 iff i is in the list of its s-th neighbour; rev lists, per row, the NON-mutual edges of the anchors with
    0 < a <= 1 that point at it.  Coordinates and labels only: part of a stage's plan; ContrastStage's backward then
    gathers every gradient row (no float atomics).  dist2: the squared distances knnquery returned with neighbor_idx (the
    same view of them) when the lists are the self-search of the stage's cloud -- membership then follows from one distance
    comparison per edge; None scans the neighbours' lists (any lists)."""
    _need_gpu(neighbor_idx, a)
    _need_dtype(torch.float32, a=a)
    nptr, k, stride, keep = _nbr_view(neighbor_idx)
    m = neighbor_idx.shape[0]
    lib = _lib.load()
    dptr = None
    if dist2 is not None:
        _need_dtype(torch.float32, dist2=dist2)
        assert dist2.shape == neighbor_idx.shape and dist2.stride() == neighbor_idx.stride() and dist2.device == a.device
        dptr = _ptr(dist2)
    mutual = torch.empty(m, k, dtype=torch.uint8, device=a.device)
    rev = torch.empty(m + 1 + m * k, dtype=torch.int32, device=a.device)
    wbytes = int(lib.amc3d_contrast_mutual_workspace_bytes(m))
    work = torch.empty(wbytes, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device), timing.span("contrast_mutual", m * k * 9 + m * 8, moved=m * k * (13 if dist2 is not None else 5 + 4 * k)):
        _lib.check(lib.amc3d_contrast_mutual(m, k, stride, nptr, dptr, _ptr(a), _ptr(mutual), _ptr(rev), _ptr(work), wbytes,
                                             _stream(a)), "contrast_mutual")
    return mutual, rev


class ContrastStage(Function):
    """Stage loss of ContrastHead.point_contrast_margin (MarginContrast.py:250-257): mean over the
    anchors with 0 < a <= 1 of the margin soft-NN loss on cosine similarities.  anchors: select_anchors(a) of
    the same a, or None (every anchor is then visited and tested)."""

    @staticmethod
    def forward(ctx, features, neighbor_idx, posmask, a, mu, nu, temperature, anchors=None, rev=None, mutual=None):
        _need_gpu(features, neighbor_idx, posmask, a)
        f = features.contiguous()
        assert f.dtype == torch.float32 and posmask.dtype == torch.bool and posmask.is_contiguous()
        if anchors is not None:
            _need_dtype(torch.int32, anchors=anchors)
            assert anchors.is_contiguous() and anchors.numel() >= f.shape[0] + 1 and anchors.device == f.device
        nptr, k, stride, keep = _nbr_view(neighbor_idx)
        m, C = f.shape
        dev = f.device
        norm = torch.empty(m, dtype=torch.float32, device=dev)
        lib = _lib.load()
        # the unit rows f_i / |f_i| the row-gather kernels read (forward, and the mutual-edge backward instead of f)
        unit = torch.empty_like(f) if lib.amc3d_contrast_backward_csr_supported(C) else None
        sim = torch.empty(m, k, dtype=torch.float32, device=dev)
        loss_pt = torch.empty(m, dtype=torch.float32, device=dev)
        mean_cnt = torch.empty(2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), timing.span("contrast_forward", m * C * 4 + m * k * 9 + m * 12, moved=m * C * 4 * (3 + k) + m * k * 9 + m * 12):
            _lib.check(lib.amc3d_contrast_forward(m, C, k, stride, _ptr(f), nptr, _ptr(posmask), _ptr(a),
                                                  _ptr(anchors) if anchors is not None else None,
                                                  float(mu), float(nu), float(temperature), _ptr(norm),
                                                  _ptr(unit) if unit is not None else None,
                                                  _ptr(sim), None, _ptr(loss_pt), _ptr(mean_cnt), _stream(f)),
                       "contrast_forward")
        if rev is not None:
            _need_dtype(torch.int32, rev=rev)
            assert anchors is not None and rev.is_contiguous() and rev.numel() == m + 1 + m * k and rev.device == dev
        if mutual is not None:  # rev then holds the non-mutual edges only (contrast_mutual)
            _need_dtype(torch.uint8, mutual=mutual)
            assert rev is not None and mutual.is_contiguous() and mutual.shape == (m, k) and mutual.device == dev
        ctx.save_for_backward(f, norm, keep, posmask, a, sim, mean_cnt, anchors, rev, mutual, unit)
        ctx.args = (float(mu), float(nu), float(temperature), k, stride)
        return mean_cnt[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        f, norm, nbr, posmask, a, sim, mean_cnt, anchors, rev, mutual, unit = ctx.saved_tensors
        mu, nu, temperature, k, stride = ctx.args
        m, C = f.shape
        g = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        nptr = _ptr(nbr)  # data_ptr includes the offset of an idx[:, 1:] view: the first used column
        lib = _lib.load()
        if mutual is not None and unit is not None:
            grad_f = torch.empty_like(f)  # every row is written
            wb = int(lib.amc3d_contrast_backward_mutual_workspace_bytes(m))
            work = torch.empty(wb, dtype=torch.uint8, device=f.device)
            with torch.cuda.device(f.device), timing.span("contrast_backward", m * C * 8 + m * k * 10 + m * 8,
                                                          moved=m * C * 4 * (2 + k) + m * k * 42 + m * 40):
                _lib.check(lib.amc3d_contrast_backward_mutual(m, C, k, stride, _ptr(unit), _ptr(norm), nptr, _ptr(posmask), _ptr(a),
                                                              _ptr(mutual), _ptr(rev), mu, nu, temperature, _ptr(sim), None,
                                                              _ptr(mean_cnt), _ptr(g), _ptr(work), wb, _ptr(grad_f), _stream(f)),
                           "contrast_backward_mutual")
            return (grad_f,) + (None,) * 9
        if mutual is not None:
            rev = None  # (a width the row kernels do not cover: the atomic form below; rev holds non-mutual edges only)
        if rev is not None and lib.amc3d_contrast_backward_csr_supported(C):
            grad_f = torch.empty_like(f)  # every row is written
            gco = torch.empty(m * k, dtype=torch.float32, device=f.device)
            with torch.cuda.device(f.device), timing.span("contrast_backward_csr", m * C * 8 + m * k * 17, moved=m * C * 4 * (2 + 2 * k) + m * k * 17):
                _lib.check(lib.amc3d_contrast_backward_csr(m, C, k, stride, _ptr(f), _ptr(norm), nptr, _ptr(posmask),
                                                           _ptr(a), _ptr(anchors), _ptr(rev), mu, nu, temperature,
                                                           _ptr(sim), _ptr(mean_cnt), _ptr(g), _ptr(gco), _ptr(grad_f),
                                                           _stream(f)), "contrast_backward_csr")
            return (grad_f,) + (None,) * 9
        grad_f = torch.zeros_like(f)
        with torch.cuda.device(f.device), timing.span("contrast_backward", m * C * 8 + m * k * 9 + m * 8, moved=m * C * 4 * (1 + 2 * k) + m * k * 9):
            _lib.check(_lib.load().amc3d_contrast_backward(m, C, k, stride, _ptr(f), _ptr(norm), nptr, _ptr(posmask),
                                                           _ptr(a), _ptr(anchors) if anchors is not None else None,
                                                           mu, nu, temperature, _ptr(sim), _ptr(mean_cnt),
                                                           _ptr(g), _ptr(grad_f), _stream(f)), "contrast_backward")
        return (grad_f,) + (None,) * 9


contrast_stage = ContrastStage.apply


class ContrastStageChannelMajor(Function):
    """contrast_stage on the decoder's channel-major embeddings f_cm (B, C, n) -- what the reference flattens into (B*n, C) rows
    first (pointnext_AA.py:518-519).  The point-major copy of f is never made: the forward writes the unit rows f_i / |f_i|
    through an LDS tile, the mutual-edge backward reads only those; its gradient rows go back through one tiled transpose.
    Needs the mutual-edge plan (anchors, rev, mutual) and C in {16, 32, 64, 128, 256}: contrast_stage_supported_cm()."""

    @staticmethod
    def forward(ctx, f_cm, neighbor_idx, posmask, a, mu, nu, temperature, anchors, rev, mutual):
        _need_gpu(f_cm, neighbor_idx, posmask, a, anchors, rev, mutual)
        _need_dtype(torch.float32, f_cm=f_cm)
        _need_dtype(torch.int32, anchors=anchors, rev=rev)
        _need_dtype(torch.uint8, mutual=mutual)
        f_cm = f_cm.contiguous()
        B, C, n = f_cm.shape
        m = B * n
        nptr, k, stride, keep = _nbr_view(neighbor_idx)
        assert posmask.dtype == torch.bool and posmask.is_contiguous() and neighbor_idx.shape[0] == m
        assert anchors.is_contiguous() and anchors.numel() >= m + 1 and rev.is_contiguous() and rev.numel() == m + 1 + m * k
        assert mutual.is_contiguous() and mutual.shape == (m, k)
        dev = f_cm.device
        norm = torch.empty(m, dtype=torch.float32, device=dev)
        unit = torch.empty(m, C, dtype=torch.float32, device=dev)
        # the two sums of exponentials per anchor, kept for the backward's records; the cosines themselves are not stored (the
        # mutual-edge backward recomputes the ones it needs from the unit rows it fetches anyway)
        stats = torch.empty(m, 2, dtype=torch.float32, device=dev)
        loss_pt = torch.empty(m, dtype=torch.float32, device=dev)
        mean_cnt = torch.empty(2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev), timing.span("contrast_forward", m * C * 4 + m * k * 5 + m * 20, moved=m * C * 4 * (2 + k) + m * k * 5 + m * 20):
            _lib.check(_lib.load().amc3d_contrast_forward_cm(B, C, n, k, stride, _ptr(f_cm), nptr, _ptr(posmask), _ptr(a), _ptr(anchors),
                                                             float(mu), float(nu), float(temperature), _ptr(norm), _ptr(unit),
                                                             None, _ptr(stats), _ptr(loss_pt), _ptr(mean_cnt), _stream(f_cm)),
                       "contrast_forward_cm")
        ctx.save_for_backward(norm, keep, posmask, a, stats, mean_cnt, rev, mutual, unit)
        ctx.args = (float(mu), float(nu), float(temperature), k, stride, B, C, n)
        return mean_cnt[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        norm, nbr, posmask, a, stats, mean_cnt, rev, mutual, unit = ctx.saved_tensors
        mu, nu, temperature, k, stride, B, C, n = ctx.args
        m = B * n
        g = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        lib = _lib.load()
        grad_rows = torch.empty_like(unit)  # every row is written
        grad_cm = torch.empty(B, C, n, dtype=torch.float32, device=unit.device)
        wb = int(lib.amc3d_contrast_backward_mutual_workspace_bytes(m))
        work = torch.empty(wb, dtype=torch.uint8, device=unit.device)
        with torch.cuda.device(unit.device), timing.span("contrast_backward", m * C * 8 + m * k * 10 + m * 8,
                                                         moved=m * C * 4 * (4 + k) + m * k * 42 + m * 40):
            _lib.check(lib.amc3d_contrast_backward_mutual(m, C, k, stride, _ptr(unit), _ptr(norm), _ptr(nbr), _ptr(posmask), _ptr(a),
                                                          _ptr(mutual), _ptr(rev), mu, nu, temperature, None, _ptr(stats),
                                                          _ptr(mean_cnt), _ptr(g), _ptr(work), wb, _ptr(grad_rows), _stream(unit)),
                       "contrast_backward_mutual")
            # (B, n, C) -> (B, C, n)
            _lib.check(lib.amc3d_transpose_cn(B, n, C, _ptr(grad_rows), _ptr(grad_cm), _stream(unit)), "transpose_cn")
        return (grad_cm,) + (None,) * 9


def contrast_stage_supported_cm(f_cm, anchors, rev, mutual):
    return (torch.is_tensor(f_cm) and f_cm.is_cuda and f_cm.dtype == torch.float32 and f_cm.dim() == 3 and anchors is not None
            and rev is not None and mutual is not None and bool(_lib.load().amc3d_contrast_backward_csr_supported(int(f_cm.shape[1]))))


contrast_stage_cm = ContrastStageChannelMajor.apply


# ----------------------------------------------------------------------------------------------
# training-mode BatchNorm fused with ReLU / neighbourhood max-pool (csrc/bn.hip)
# ----------------------------------------------------------------------------------------------
def _bn_ws(C, dev, extra=0):
    n = int(_lib.load().amc3d_bn_workspace_bytes(C)) + extra
    return torch.empty(n, dtype=torch.uint8, device=dev), n


def _bn_running_args(bn):
    """(momentum, running_mean ptr, running_var ptr, num_batches_tracked ptr) of an nn.BatchNorm module whose
    buffers the fused forward updates in place, or (0, None, None, None)"""
    if bn is None or not bn.track_running_stats or bn.running_mean is None:
        return 0.0, None, None, None
    assert bn.running_mean.dtype == torch.float32 and bn.num_batches_tracked.dtype == torch.int64
    mom = -1.0 if bn.momentum is None else float(bn.momentum)
    return mom, _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(bn.num_batches_tracked)


class BatchNormAct(Function):
    """y = [relu](batch_norm(x)) with batch statistics; x (B, C, *) contiguous fp32.  `bn`: the nn.BatchNorm module
    whose running buffers are updated in the same launch (None: no update).
    Returns (y, batch mean, unbiased batch variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, bn=None):
        _need_gpu(x, gamma, beta)
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        y = torch.empty_like(x)
        work, wb = _bn_ws(C, dev)
        lib = _lib.load()
        mom, rm, rv, nbt = _bn_running_args(bn)
        with torch.cuda.device(dev), timing.span("bn_act_forward", x.numel() * 8, moved=x.numel() * 12):
            _lib.check(lib.amc3d_bn_forward(B, C, L, 0, int(bool(relu)), float(eps), mom, _ptr(x), _ptr(gamma), _ptr(beta),
                                            _ptr(y), None, _ptr(mean), _ptr(invstd), _ptr(var_u), rm, rv, nbt,
                                            _ptr(work), wb, _stream(x)), "bn_forward")
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = bool(relu)
        ctx.mark_non_differentiable(mean, var_u)
        ctx.set_materialize_grads(False)  # no zero tensors for the statistics' (absent) gradients
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        work, wb = _bn_ws(C, x.device, extra=C * 8)
        with torch.cuda.device(x.device), timing.span("bn_act_backward", x.numel() * 12, moved=x.numel() * 20):
            _lib.check(_lib.load().amc3d_bn_backward(B, C, L, 1, int(ctx.relu), _ptr(x), _ptr(dy), None, _ptr(mean),
                                                     _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dx), _ptr(dgamma),
                                                     _ptr(dbeta), _ptr(work), wb, _stream(x)), "bn_backward")
        return dx, dgamma, dbeta, None, None, None


class BatchNormResidualAct(Function):
    """y = relu(batch_norm(x) + res) with batch statistics: the tail of an InvResMLP block (pointnext_AA.py:296-307: the
    last Conv1d -> BatchNorm1d of pwconv, `f += identity`, `self.act(f)`) in the two launches of a plain BatchNorm layer;
    backward: dres = dy * (y > 0) and the BatchNorm backward of that, two launches.  Returns (y, mean, unbiased variance)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, bn=None):
        _need_gpu(x, res, gamma, beta)
        _need_dtype(torch.float32, x=x, res=res, gamma=gamma, beta=beta)
        x, res = x.contiguous(), res.contiguous()
        assert x.shape == res.shape
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        y = torch.empty_like(x)
        work, wb = _bn_ws(C, dev)
        mom, rm, rv, nbt = _bn_running_args(bn)
        with torch.cuda.device(dev), timing.span("bn_residual_forward", x.numel() * 12, moved=x.numel() * 16):
            _lib.check(_lib.load().amc3d_bn_residual_forward(B, C, L, float(eps), mom, _ptr(x), _ptr(res), _ptr(gamma), _ptr(beta),
                                                             _ptr(y), _ptr(mean), _ptr(invstd), _ptr(var_u), rm, rv, nbt,
                                                             _ptr(work), wb, _stream(x)), "bn_residual_forward")
        ctx.save_for_backward(x, y, gamma, beta, mean, invstd)
        ctx.mark_non_differentiable(mean, var_u)
        ctx.set_materialize_grads(False)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        work, wb = _bn_ws(C, x.device, extra=C * 8)
        with torch.cuda.device(x.device), timing.span("bn_residual_backward", x.numel() * 20, moved=x.numel() * 28):
            _lib.check(_lib.load().amc3d_bn_residual_backward(B, C, L, _ptr(x), _ptr(y), _ptr(dy), _ptr(mean), _ptr(invstd),
                                                              _ptr(gamma), _ptr(beta), _ptr(dx), _ptr(dres), _ptr(dgamma),
                                                              _ptr(dbeta), _ptr(work), wb, _stream(x)), "bn_residual_backward")
        return dx, dres, dgamma, dbeta, None, None


class BatchNormSigmoid(Function):
    """y = sigmoid(batch_norm(x)) with batch statistics: nn.BatchNorm1d -> nn.Sigmoid of the APM towers of AMContrast3D++
    (openpoints/AMContrast3D/APM/concatenation.py:20-60) in the two launches of a plain BatchNorm layer; backward from the saved
    output (dy * y (1 - y), then BatchNorm backward), two launches.  Returns (y, mean, unbiased variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, bn=None):
        _need_gpu(x, gamma, beta)
        _need_dtype(torch.float32, x=x, gamma=gamma, beta=beta)
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        y = torch.empty_like(x)
        work, wb = _bn_ws(C, dev)
        mom, rm, rv, nbt = _bn_running_args(bn)
        with torch.cuda.device(dev), timing.span("bn_sigmoid_forward", x.numel() * 8, moved=x.numel() * 12):
            _lib.check(_lib.load().amc3d_bn_sigmoid_forward(B, C, L, float(eps), mom, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y),
                                                            _ptr(mean), _ptr(invstd), _ptr(var_u), rm, rv, nbt, _ptr(work), wb,
                                                            _stream(x)), "bn_sigmoid_forward")
        ctx.save_for_backward(x, y, gamma, beta, mean, invstd)
        ctx.mark_non_differentiable(mean, var_u)
        ctx.set_materialize_grads(False)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        work, wb = _bn_ws(C, x.device, extra=C * 8)
        with torch.cuda.device(x.device), timing.span("bn_sigmoid_backward", x.numel() * 16, moved=x.numel() * 24):
            _lib.check(_lib.load().amc3d_bn_sigmoid_backward(B, C, L, _ptr(x), _ptr(y), _ptr(dy), _ptr(mean), _ptr(invstd),
                                                             _ptr(gamma), _ptr(beta), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                                                             _ptr(work), wb, _stream(x)), "bn_sigmoid_backward")
        return dx, dgamma, dbeta, None, None


@torch.no_grad()
def bn_update_running(bn, mean, var_unbiased):
    """nn.BatchNorm's training-mode buffer update (num_batches_tracked, running_mean, running_var) in one launch."""
    _need_gpu(mean, var_unbiased, bn.running_mean, bn.running_var, bn.num_batches_tracked)
    mom = -1.0 if bn.momentum is None else float(bn.momentum)
    _lib.check(_lib.load().amc3d_bn_update_running(mean.numel(), mom, _ptr(mean), _ptr(var_unbiased),
                                                   _ptr(bn.running_mean), _ptr(bn.running_var),
                                                   _ptr(bn.num_batches_tracked), _stream(mean)), "bn_update_running")


class BatchNormMax(Function):
    """y (B,C,M) = max over the K neighbours of [relu](batch_norm(x (B,C,M,K))) with batch statistics.
    Returns (y, batch mean, unbiased batch variance); the (B,C,M,K) normalised tensor is never written."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, bn=None):
        _need_gpu(x, gamma, beta)
        x = x.contiguous()
        B, C, M, K = x.shape
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        y = torch.empty(B, C, M, dtype=torch.float32, device=dev)
        arg = torch.empty(B, C, M, dtype=torch.uint8, device=dev)
        work, wb = _bn_ws(C, dev)
        lib = _lib.load()
        mom, rm, rv, nbt = _bn_running_args(bn)
        with torch.cuda.device(dev), timing.span("bn_max_forward", x.numel() * 4 + y.numel() * 5, moved=x.numel() * 8 + y.numel() * 5):
            _lib.check(lib.amc3d_bn_forward(B, C, M * K, K, int(bool(relu)), float(eps), mom, _ptr(x), _ptr(gamma),
                                            _ptr(beta), _ptr(y), _ptr(arg), _ptr(mean), _ptr(invstd), _ptr(var_u), rm, rv,
                                            nbt, _ptr(work), wb, _stream(x)), "bn_forward")
        ctx.save_for_backward(x, gamma, beta, mean, invstd, arg)
        ctx.relu = bool(relu)
        if _pool_log is not None:
            _pool_log[_next_pool_seq()] = arg
        ctx.mark_non_differentiable(mean, var_u)
        ctx.set_materialize_grads(False)  # no zero tensors for the statistics' (absent) gradients
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, gamma, beta, mean, invstd, arg = ctx.saved_tensors
        B, C, M, K = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        work, wb = _bn_ws(C, x.device, extra=C * 8)
        with torch.cuda.device(x.device), timing.span("bn_max_backward", x.numel() * 4 + dy.numel() * 9, moved=x.numel() * 8 + dy.numel() * 5):
            _lib.check(_lib.load().amc3d_bn_backward(B, C, M * K, K, int(ctx.relu), _ptr(x), _ptr(dy), _ptr(arg),
                                                     _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dx),
                                                     _ptr(dgamma), _ptr(dbeta), _ptr(work), wb, _stream(x)),
                       "bn_backward")
        return dx, dgamma, dbeta, None, None, None


@torch.no_grad()
def bn_eval(x, bn, relu, pool=False):
    """Inference-mode BatchNorm (running statistics) [+ReLU] [+max over the last, neighbour, dimension] in one pass:
    the whole-room evaluation loop of the reference (examples/segmentation/main_AA.py:431-802) runs the same blocks with
    model.eval().  y = ((x - running_mean) * rsqrt(running_var + eps)) * weight + bias, no gradient."""
    _need_gpu(x, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    x = x.contiguous()
    B, C = x.shape[0], x.shape[1]
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        if pool:
            M, K = x.shape[-2], x.shape[-1]
            y = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
            arg = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
            _lib.check(lib.amc3d_bn_max(B, C, M, K, int(bool(relu)), _ptr(x), _ptr(bn.running_mean), _ptr(invstd),
                                        _ptr(bn.weight), _ptr(bn.bias), _ptr(y), _ptr(arg), _stream(x)), "bn_max")
            return y
        L = x.numel() // (B * C)
        y = torch.empty_like(x)
        _lib.check(lib.amc3d_bn_act(B, C, L, int(bool(relu)), _ptr(x), _ptr(bn.running_mean), _ptr(invstd),
                                    _ptr(bn.weight), _ptr(bn.bias), _ptr(y), _stream(x)), "bn_act")
        return y


class SyncBatchNormFused(Function):
    """BatchNormAct (pool=False) / BatchNormMax (pool=True) with statistics over every rank of `group`:
    torch.nn.SyncBatchNorm's arithmetic (torch/nn/modules/_functions.py; the reference converts all BN layers to it
    when world_size > 1, examples/segmentation/main_AA.py:146-148) on the fused kernels.  Per layer one all-reduce of
    2C+1 doubles forward ({sum x, sum x^2} per channel and the element count) and one of 2C doubles backward
    ({sum dq, sum dq*xhat}); parameter gradients stay rank-local, as torch's do.  Counts stay on the device, so ranks
    may hold different numbers of points and the whole sequence can be captured in a graph."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, pool, bn, group):
        import torch.distributed as dist
        _need_gpu(x, gamma, beta)
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        K = x.shape[-1] if pool else 0
        dev = x.device
        lib = _lib.load()
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        sums[2 * C:].fill_(float(B * L))
        work, wb = _bn_ws(C, dev)
        with torch.cuda.device(dev):
            _lib.check(lib.amc3d_bn_sums(B, C, L, _ptr(x), _ptr(sums), _ptr(work), wb, _stream(x)), "bn_sums")
        from . import graphs
        graphs.collective(lambda: dist.all_reduce(sums, group=group))  # eager, between two captured segments (graphs.py)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        var_u = torch.empty_like(mean)
        if pool:
            y = torch.empty(x.shape[:-1], dtype=torch.float32, device=dev)
            arg = torch.empty(x.shape[:-1], dtype=torch.uint8, device=dev)
        else:
            y, arg = torch.empty_like(x), None
        mom, rm, rv, nbt = _bn_running_args(bn)
        with torch.cuda.device(dev):
            _lib.check(lib.amc3d_bn_forward_synced(B, C, L, K, int(bool(relu)), float(eps), mom, _ptr(x), _ptr(sums),
                                                   _ptr(gamma), _ptr(beta), _ptr(y), None if arg is None else _ptr(arg),
                                                   _ptr(mean), _ptr(invstd), _ptr(var_u), rm, rv, nbt, _stream(x)),
                       "bn_forward_synced")
        saved = [x, gamma, beta, mean, invstd, sums] + ([arg] if pool else [])
        ctx.save_for_backward(*saved)
        ctx.relu, ctx.pool, ctx.group = bool(relu), bool(pool), group
        ctx.mark_non_differentiable(mean, var_u)
        ctx.set_materialize_grads(False)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        import torch.distributed as dist
        x, gamma, beta, mean, invstd, sums = ctx.saved_tensors[:6]
        arg = ctx.saved_tensors[6] if ctx.pool else None
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        K = x.shape[-1] if ctx.pool else 1
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        dsums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        work, wb = _bn_ws(C, x.device)
        lib = _lib.load()
        aptr = None if arg is None else _ptr(arg)
        with torch.cuda.device(x.device):
            _lib.check(lib.amc3d_bn_backward_sums(B, C, L, K, int(ctx.relu), _ptr(x), _ptr(dy), aptr, _ptr(mean),
                                                  _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dsums), _ptr(dgamma),
                                                  _ptr(dbeta), _ptr(work), wb, _stream(x)), "bn_backward_sums")
        from . import graphs
        group = ctx.group
        graphs.collective(lambda: dist.all_reduce(dsums, group=group))
        with torch.cuda.device(x.device):
            _lib.check(lib.amc3d_bn_backward_synced(B, C, L, K, int(ctx.relu), _ptr(x), _ptr(dy), aptr, _ptr(mean),
                                                    _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dsums),
                                                    ctypes.c_void_p(sums.data_ptr() + 16 * C), _ptr(dx), _stream(x)),
                       "bn_backward_synced")
        return dx, dgamma, dbeta, None, None, None, None, None


# ----------------------------------------------------------------------------------------------
# grouped 1x1 convolution fused with its gather, fp32 MFMA (csrc/gcc.hip)
# ----------------------------------------------------------------------------------------------
def grouped_conv_supported(cin, cout):
    return bool(_lib.load().amc3d_grouped_conv_supported(int(cin), int(cout)))


class GroupedConv(Function):
    """y (B,Cout,M,K) = W . [dp ; features[:, :, idx]] -- grouping_operation + torch.cat + nn.Conv2d 1x1 of the
    reference (group.py:244-255,323-325; pointnext_AA.py:164-166) without the (B,C+3,M,K) tensor.
    features (B,C,N) fp32, dp (B,3,M,K), idx (B,M,K) int32, weight (Cout,C+3,1,1)."""

    @staticmethod
    def forward(ctx, features, dp, idx, weight):
        _need_gpu(features, dp, idx, weight)
        features, dp, idx = features.contiguous(), dp.contiguous(), idx.contiguous()
        B, C, N = features.shape
        _, M, K = idx.shape
        Cout = weight.shape[0]
        assert weight.shape[1] == C + 3 and idx.dtype == torch.int32
        dev = features.device
        f_pm = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        y = torch.empty(B, Cout, M, K, dtype=torch.float32, device=dev)
        w2 = weight.reshape(Cout, C + 3).contiguous()
        lib = _lib.load()
        flops = 2.0 * B * M * K * (C + 3) * Cout
        with torch.cuda.device(dev), timing.span("grouped_conv_forward", B * M * K * (4 * C + 16 + 4 * Cout), flops):
            _lib.check(lib.amc3d_transpose_cn(B, C, N, _ptr(features), _ptr(f_pm), _stream(features)), "transpose_cn")
            _lib.check(lib.amc3d_grouped_conv_forward(B, C, Cout, N, M, K, _ptr(f_pm), _ptr(dp), _ptr(idx), _ptr(w2),
                                                      _ptr(y), _stream(features)), "grouped_conv_forward")
        ctx.save_for_backward(f_pm, dp, idx, w2)
        ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        f_pm, dp, idx, w2 = ctx.saved_tensors
        B, N, C = f_pm.shape
        _, M, K = idx.shape
        Cout = w2.shape[0]
        dy = dy.contiguous()
        dev = dy.device
        need_f, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        df_pm = torch.zeros(B, N, C, dtype=torch.float32, device=dev) if need_f else None
        dw = torch.empty(Cout, C + 3, dtype=torch.float32, device=dev) if need_w else None
        lib = _lib.load()
        wb = int(lib.amc3d_grouped_conv_workspace_bytes(B, C, Cout, M, K))
        work = torch.empty(max(wb, 4), dtype=torch.uint8, device=dev)
        flops = 2.0 * B * M * K * Cout * ((C if need_f else 0) + (C + 3 if need_w else 0))
        with torch.cuda.device(dev), timing.span("grouped_conv_backward", B * M * K * (8 * C + 16 + 8 * Cout), flops):
            _lib.check(lib.amc3d_grouped_conv_backward(B, C, Cout, N, M, K, _ptr(f_pm), _ptr(dp), _ptr(idx), _ptr(w2),
                                                       _ptr(dy), _ptr(df_pm) if need_f else None,
                                                       _ptr(dw) if need_w else None, _ptr(work), wb, _stream(dy)),
                       "grouped_conv_backward")
        df = df_pm.transpose(1, 2).contiguous() if need_f else None
        return df, None, None, (dw.view(ctx.wshape) if need_w else None)


grouped_conv = GroupedConv.apply


class PointMajorRows(Function):
    """(B,C,n) channel-major features -> (B*n, C) rows: torch.flatten(f.transpose(1, 2), 0, 1), the per-stage embedding
    the contrastive loss reads (pointnext_AA.py:459-460, 518-519), as ONE coalesced LDS-tiled transpose each way.  torch
    makes the copy with a strided elementwise kernel and, in backward, adds the strided gradient view into the
    channel-major gradient with another one (0.7 GB of scattered traffic per step, measured)."""

    @staticmethod
    def forward(ctx, f):
        _need_gpu(f)
        _need_dtype(torch.float32, f=f)
        f = f.contiguous()
        B, C, n = f.shape
        out = torch.empty(B * n, C, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device), timing.span("transpose_rows", f.numel() * 8):
            _lib.check(_lib.load().amc3d_transpose_cn(B, C, n, _ptr(f), _ptr(out), _stream(f)), "transpose_cn")
        ctx.shape = (B, C, n)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, n = ctx.shape
        g = g.contiguous()
        out = torch.empty(B, C, n, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device), timing.span("transpose_rows", g.numel() * 8):
            # (B, n, C) -> (B, C, n): the same kernel with the roles of the two axes exchanged
            _lib.check(_lib.load().amc3d_transpose_cn(B, n, C, _ptr(g), _ptr(out), _stream(g)), "transpose_cn")
        return out


def point_major_rows(f):
    """(B,C,n) -> (B*n,C); the fused transpose on fp32 GPU tensors, torch's flatten(transpose) otherwise"""
    import os
    if f.is_cuda and f.dtype == torch.float32 and f.dim() == 3 and not os.environ.get("AMC3D_NO_PM_ROWS"):
        return PointMajorRows.apply(f)
    return torch.flatten(f.transpose(1, 2), start_dim=0, end_dim=1)


# ----------------------------------------------------------------------------------------------
# single grouped conv + BatchNorm [+ReLU] + max over the neighbours, convolved before the gather (csrc/lagg.hip)
# ----------------------------------------------------------------------------------------------
def local_aggregation_supported(cout, nsample):
    return bool(_lib.load().amc3d_local_aggregation_supported(int(cout), int(nsample)))


@torch.no_grad()
def group_moments(idx, dp, n_support):
    """Geometry moments of a neighbourhood query -- idx (B,M,K) int32 into n_support points, dp (B,3,M,K) -- that the
    convolve-before-gather layers need for their BatchNorm statistics and backward (in-degree and dp sum per support point,
    global dp moments): coordinates only, so part of the geometry plan.  -> opaque uint8 buffer."""
    _need_gpu(idx, dp)
    _need_dtype(torch.int32, idx=idx)
    _need_dtype(torch.float32, dp=dp)
    idx, dp = idx.contiguous(), dp.contiguous()
    B, M, K = idx.shape
    lib = _lib.load()
    nb = int(lib.amc3d_group_moments_bytes(B, int(n_support)))
    out = torch.empty(nb, dtype=torch.uint8, device=idx.device)
    with torch.cuda.device(idx.device), timing.span("group_moments", idx.numel() * 16 + nb):
        _lib.check(lib.amc3d_group_moments(B, int(n_support), M, K, _ptr(idx), _ptr(dp), _ptr(out), nb, _stream(idx)),
                   "group_moments")
    return out


@torch.no_grad()
def group_csr(idx, n_support):
    """Reverse adjacency of a neighbourhood query idx (B,M,K) into n_support points: (rev_start (B*n+1), rev_edge (B*M*K))
    int32 -- for every source point the positions that gathered it, in ascending position order (csrc/csr.hip).
    Coordinates only: part of the geometry plan.  The backward passes that used float atomics gather over these lists."""
    _need_gpu(idx)
    _need_dtype(torch.int32, idx=idx)
    idx = idx.contiguous()
    B, M, K = idx.shape
    n = int(n_support)
    dev = idx.device
    lib = _lib.load()
    rev_start = torch.empty(B * n + 1, dtype=torch.int32, device=dev)
    rev_edge = torch.empty(B * M * K, dtype=torch.int32, device=dev)
    wb = int(lib.amc3d_group_csr_workspace_bytes(B, M, K))
    work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), timing.span("group_csr", idx.numel() * 12 + B * n * 4):
        _lib.check(lib.amc3d_group_csr(B, n, M, K, _ptr(idx), _ptr(rev_start), _ptr(rev_edge), _ptr(work), wb, _stream(idx)),
                   "group_csr")
    return rev_start, rev_edge


@torch.no_grad()
@torch.no_grad()
def group_csr_dp(idx, dp, rev_edge):
    """(dp, position) of every edge of group_csr's lists in list order, (B*M*K, 4) fp32 (the position as int32 bits in the
    fourth float): the stream the gathering backward of GroupedConvBN reads instead of an edge id + three scattered floats"""
    _need_gpu(idx, dp, rev_edge)
    _need_dtype(torch.float32, dp=dp)
    _need_dtype(torch.int32, rev_edge=rev_edge)
    B, M, K = idx.shape
    dp = dp.contiguous()
    assert dp.numel() == B * 3 * M * K and rev_edge.numel() == B * M * K and rev_edge.is_contiguous()
    out = torch.empty(B * M * K, 4, dtype=torch.float32, device=idx.device)
    with torch.cuda.device(idx.device):
        _lib.check(_lib.load().amc3d_group_csr_dp(B, M, K, _ptr(rev_edge), _ptr(dp), _ptr(out), _stream(idx)), "group_csr_dp")
    return out


def group_moments_csr(idx, dp, n_support, csr):
    """group_moments from the reverse lists: exact in-degree, dp sums in list order, no scattered atomics"""
    _need_gpu(idx, dp)
    dp = dp.contiguous()
    B, M, K = idx.shape
    lib = _lib.load()
    nb = int(lib.amc3d_group_moments_bytes(B, int(n_support)))
    out = torch.empty(nb, dtype=torch.uint8, device=idx.device)
    with torch.cuda.device(idx.device), timing.span("group_moments", idx.numel() * 16 + nb):
        _lib.check(lib.amc3d_group_moments_csr(B, int(n_support), M, K, _ptr(csr[0]), _ptr(csr[1]), _ptr(dp), _ptr(out), nb,
                                               _stream(idx)), "group_moments_csr")
    return out


def _sync(group, buf):
    """all-reduce of a statistics buffer between the two phases of a layer: eager, between captured graph segments"""
    import torch.distributed as dist
    from . import graphs
    graphs.collective(lambda: dist.all_reduce(buf, group=group))


class LocalAggregationFused(Function):
    """pooled (B,C,M) = max_k [relu](bn(conv1x1([dp ; f[idx]]))) -- grouping_operation + cat + Conv2d + BatchNorm2d (batch
    statistics) [+ ReLU] + max of LocalAggregation / single-layer SetAbstraction (pointnext_AA.py:57-63, 139-170) -- with
    the conv applied to the N source points BEFORE the gather:  W.[dp ; f[idx]] = (W_f.f)[idx] + W_dp.dp.
    f (B,Cin,N) fp32, dp (B,3,M,K), idx (B,M,K) int32, moments = group_moments(idx, dp, N), weight (C,Cin+3,1,1),
    `bn`: the nn.BatchNorm2d whose running buffers are updated (None: no update); `group`: a process group over which the
    statistics are taken (nn.SyncBatchNorm semantics: one all-reduce of 2C+1 doubles forward, 2C backward), or None."""

    @staticmethod
    def forward(ctx, f, dp, idx, moments, weight, gamma, beta, eps, relu, bn=None, group=None):
        _need_gpu(f, dp, idx, moments, weight, gamma, beta)
        _need_dtype(torch.float32, f=f, dp=dp, weight=weight)
        _need_dtype(torch.int32, idx=idx)
        f, dp, idx = f.contiguous(), dp.contiguous(), idx.contiguous()
        B, Cin, N = f.shape
        _, M, K = idx.shape
        C = weight.shape[0]
        assert weight.numel() == C * (Cin + 3)
        dev = f.device
        lib = _lib.load()
        w2 = weight.reshape(C, Cin + 3)
        w_dp, w_f = _split_columns(w2, 3)
        g_cm = torch.empty(B, C, N, dtype=torch.float32, device=dev)
        g_pm = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        pooled = torch.empty(B, C, M, dtype=torch.float32, device=dev)
        ystar = torch.empty_like(pooled)
        arg = torch.empty(B, C, M, dtype=torch.uint8, device=dev)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd, var_u = torch.empty_like(mean), torch.empty_like(mean)
        gd = torch.empty(C, 3, dtype=torch.float64, device=dev)
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        wb = int(lib.amc3d_local_aggregation_workspace_bytes(B, C, N, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        mom, rm, rv, nbt = _bn_running_args(bn)
        ctx.bf16 = mixed_precision() and min(Cin, C) >= 64 and B * N >= 4096  # the conv on the source points on the bf16 MFMA

        def call(phase):
            _lib.check(lib.amc3d_local_aggregation_forward(
                B, C, N, M, K, 1, int(bool(relu)), float(eps), mom, _ptr(g_cm), _ptr(idx), _ptr(dp), _ptr(w_dp),
                _ptr(moments), _ptr(gamma), _ptr(beta), _ptr(g_pm), _ptr(pooled), _ptr(arg), _ptr(ystar), _ptr(mean),
                _ptr(invstd), _ptr(var_u), _ptr(gd), rm, rv, nbt, phase, _ptr(sums), _ptr(work), wb, _stream(f)),
                "local_aggregation_forward")

        with torch.cuda.device(dev):
            with timing.span("pointwise_conv_forward", 4 * B * N * (Cin + C), 2.0 * B * N * Cin * C):
                _pw_forward(lib, ctx.bf16, B, Cin, C, N, f, w_f, None, g_cm)
            # algorithmic bytes: G read (statistics) + the gathered rows, idx, dp + the pooled outputs
            with timing.span("local_aggregation_forward", 4 * B * N * C + 16 * B * M * K + 5 * B * M * C,
                             moved=8 * B * N * C + B * M * K * (4 * C + 16) + 9 * B * M * C):
                if group is None:
                    call(0)
                else:
                    call(1)
                    _sync(group, sums)
                    call(2)
        if bn is not None and bn.track_running_stats and bn.running_mean is not None and bn.momentum is None:
            bn_update_running(bn, mean, var_u)  # cumulative average: its own launch
        ctx.save_for_backward(f, w_f, w_dp, g_pm, idx, dp, moments, gamma, beta, mean, invstd, gd, ystar, arg, sums)
        ctx.relu, ctx.wshape, ctx.group = bool(relu), tuple(weight.shape), group
        if _pool_log is not None:
            _pool_log[_next_pool_seq()] = arg
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        f, w_f, w_dp, g_pm, idx, dp, moments, gamma, beta, mean, invstd, gd, ystar, arg, sums = ctx.saved_tensors
        B, Cin, N = f.shape
        _, M, K = idx.shape
        C = w_f.shape[0]
        dev = f.device
        lib = _lib.load()
        dpooled = dpooled.contiguous()
        dg_cm = torch.empty(B, C, N, dtype=torch.float32, device=dev)
        dw_dp = torch.empty(C, 3, dtype=torch.float32, device=dev)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        dsums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        count = ctypes.c_void_p(sums.data_ptr() + 16 * C)
        wb = int(lib.amc3d_local_aggregation_workspace_bytes(B, C, N, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        need_f = ctx.needs_input_grad[0]
        df = torch.empty_like(f) if need_f else None
        dw_f = torch.empty(C, Cin, dtype=torch.float32, device=dev)
        _, pw_wbytes, pw_bwd = _pw(lib, ctx.bf16)
        wb2 = int(pw_wbytes(B, Cin, C, N))
        work2 = torch.empty(max(wb2, 4), dtype=torch.uint8, device=dev)

        def call(phase):
            _lib.check(lib.amc3d_local_aggregation_backward(
                B, C, N, M, K, int(ctx.relu), _ptr(dpooled), _ptr(ystar), _ptr(arg), _ptr(g_pm), _ptr(idx), _ptr(dp),
                _ptr(w_dp), _ptr(moments), _ptr(gd), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dg_cm),
                _ptr(dw_dp), _ptr(dgamma), _ptr(dbeta), phase, _ptr(dsums), count, _ptr(work), wb, _stream(f)),
                "local_aggregation_backward")

        with torch.cuda.device(dev):
            with timing.span("local_aggregation_backward", 5 * B * M * C + 8 * B * N * C + 16 * B * M * K, moved=17 * B * M * C + 12 * B * N * C + 16 * B * M * K):
                if ctx.group is None:
                    call(0)
                else:
                    call(1)
                    _sync(ctx.group, dsums)
                    call(2)
            with timing.span("pointwise_conv_backward", 4 * B * N * (Cin + C + Cin * int(need_f)),
                             2.0 * B * N * Cin * C * (1 + int(need_f)), moved=4 * B * N * (Cin + C) * (1 + int(need_f))):
                _lib.check(pw_bwd(B, Cin, C, N, _ptr(f), _ptr(w_f), _ptr(dg_cm), _ptr(df) if need_f else None, _ptr(dw_f),
                                  _ptr(work2), wb2, _stream(f)), "pointwise_conv_backward")
        dw = _join_columns(dw_dp, dw_f).view(ctx.wshape)
        return df, None, None, None, dw, dgamma, dbeta, None, None, None, None


class GroupedConvBN(Function):
    """x1 (B,C,M,32) = [relu](bn(conv1x1([dp ; f[idx]]))) -- the FIRST block of a multi-layer SetAbstraction MLP
    (PointNeXt-S: sa_layers = 2; pointnext_AA.py:104-127, 164-166) convolved before the gather like LocalAggregationFused,
    but with the activation materialised for the blocks that follow.  Backward: one pass over dx1 (csrc/lagg.hip
    lagg_collapse_kernel, or csrc/csr.hip over reverse edge lists when `csr` is given) instead of BatchNorm-backward
    statistics + apply + the conv's two backward products.  `group`: statistics over a process group (SyncBatchNorm)."""

    @staticmethod
    def forward(ctx, f, dp, idx, moments, weight, gamma, beta, eps, relu, bn=None, csr=None, group=None):
        _need_gpu(f, dp, idx, moments, weight, gamma, beta)
        _need_dtype(torch.float32, f=f, dp=dp, weight=weight)
        _need_dtype(torch.int32, idx=idx)
        f, dp, idx = f.contiguous(), dp.contiguous(), idx.contiguous()
        B, Cin, N = f.shape
        _, M, K = idx.shape
        C = weight.shape[0]
        assert weight.numel() == C * (Cin + 3)
        dev = f.device
        lib = _lib.load()
        ctx.csr = csr  # (rev_start, rev_edge[, group_csr_dp's stream]) of ops.group_csr, or None: backward then scatters with float atomics
        w2 = weight.reshape(C, Cin + 3)
        w_dp, w_f = _split_columns(w2, 3)
        g_cm = torch.empty(B, C, N, dtype=torch.float32, device=dev)
        g_pm = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        x1 = torch.empty(B, C, M, K, dtype=torch.float32, device=dev)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd, var_u = torch.empty_like(mean), torch.empty_like(mean)
        gd = torch.empty(C, 3, dtype=torch.float64, device=dev)
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        wb = int(lib.amc3d_local_aggregation_workspace_bytes(B, C, N, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        mom, rm, rv, nbt = _bn_running_args(bn)
        ctx.bf16 = mixed_precision() and min(Cin, C) >= 64 and B * N >= 4096

        def call(phase):
            _lib.check(lib.amc3d_grouped_conv_bn_forward(
                B, C, N, M, K, 1, int(bool(relu)), float(eps), mom, _ptr(g_cm), _ptr(idx), _ptr(dp), _ptr(w_dp),
                _ptr(moments), _ptr(gamma), _ptr(beta), _ptr(g_pm), _ptr(x1), _ptr(mean), _ptr(invstd), _ptr(var_u),
                _ptr(gd), rm, rv, nbt, phase, _ptr(sums), _ptr(work), wb, _stream(f)), "grouped_conv_bn_forward")

        with torch.cuda.device(dev):
            with timing.span("pointwise_conv_forward", 4 * B * N * (Cin + C), 2.0 * B * N * Cin * C):
                _pw_forward(lib, ctx.bf16, B, Cin, C, N, f, w_f, None, g_cm)
            with timing.span("grouped_conv_bn_forward", 4 * B * N * C + B * M * K * (4 * C + 16), moved=8 * B * N * C + B * M * K * (8 * C + 16)):
                if group is None:
                    call(0)
                else:
                    call(1)
                    _sync(group, sums)
                    call(2)
        if bn is not None and bn.track_running_stats and bn.running_mean is not None and bn.momentum is None:
            bn_update_running(bn, mean, var_u)
        ctx.save_for_backward(f, w_f, w_dp, g_pm, idx, dp, moments, gamma, beta, mean, invstd, gd, sums)
        ctx.relu, ctx.wshape, ctx.group = bool(relu), tuple(weight.shape), group
        return x1

    @staticmethod
    def backward(ctx, dx1):
        f, w_f, w_dp, g_pm, idx, dp, moments, gamma, beta, mean, invstd, gd, sums = ctx.saved_tensors
        B, Cin, N = f.shape
        _, M, K = idx.shape
        C = w_f.shape[0]
        dev = f.device
        lib = _lib.load()
        # SATailActivated hands the gradient over as position-major rows (a (B,C,M,K) view of a (B,M,K,C) buffer): the
        # gather along the reverse lists reads those directly; any other layout is made channel-major and transposed
        dx1_pm = ctx.csr is not None and dx1.dim() == 4 and dx1.permute(0, 2, 3, 1).is_contiguous() and C > 1
        if not dx1_pm:
            dx1 = dx1.contiguous()
        dg_cm = torch.empty(B, C, N, dtype=torch.float32, device=dev)
        dw_dp = torch.empty(C, 3, dtype=torch.float32, device=dev)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        dsums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        count = ctypes.c_void_p(sums.data_ptr() + 16 * C)
        csr = ctx.csr
        wb = int(lib.amc3d_grouped_conv_bn_csr_workspace_bytes(B, C, N, M, K) if csr is not None
                 else lib.amc3d_local_aggregation_workspace_bytes(B, C, N, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        need_f = ctx.needs_input_grad[0]
        df = torch.empty_like(f) if need_f else None
        dw_f = torch.empty(C, Cin, dtype=torch.float32, device=dev)
        _, pw_wbytes, pw_bwd = _pw(lib, ctx.bf16)
        wb2 = int(pw_wbytes(B, Cin, C, N))
        work2 = torch.empty(max(wb2, 4), dtype=torch.uint8, device=dev)

        def call(phase):
            if csr is not None:
                _lib.check(lib.amc3d_grouped_conv_bn_backward_csr(
                    B, C, N, M, K, int(ctx.relu), _ptr(dx1), int(dx1_pm), _ptr(g_pm), _ptr(csr[0]), _ptr(csr[1]),
                    _ptr(csr[2]) if len(csr) > 2 and csr[2] is not None else None, _ptr(dp), _ptr(w_dp),
                    _ptr(moments), _ptr(gd), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dg_cm), _ptr(dw_dp),
                    _ptr(dgamma), _ptr(dbeta), phase, _ptr(dsums), count, _ptr(work), wb, _stream(f)),
                    "grouped_conv_bn_backward_csr")
            else:
                _lib.check(lib.amc3d_grouped_conv_bn_backward(
                    B, C, N, M, K, int(ctx.relu), _ptr(dx1), _ptr(g_pm), _ptr(idx), _ptr(dp), _ptr(w_dp), _ptr(moments),
                    _ptr(gd), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), _ptr(dg_cm), _ptr(dw_dp), _ptr(dgamma),
                    _ptr(dbeta), phase, _ptr(dsums), count, _ptr(work), wb, _stream(f)), "grouped_conv_bn_backward")

        with torch.cuda.device(dev):
            with timing.span("grouped_conv_bn_backward", B * M * K * (8 * C + 16) + 4 * B * N * C, moved=B * M * K * (8 * C + 16) + 12 * B * N * C):
                if ctx.group is None:
                    call(0)
                else:
                    call(1)
                    _sync(ctx.group, dsums)
                    call(2)
            with timing.span("pointwise_conv_backward", 4 * B * N * (Cin + C + Cin * int(need_f)),
                             2.0 * B * N * Cin * C * (1 + int(need_f)), moved=4 * B * N * (Cin + C) * (1 + int(need_f))):
                _lib.check(pw_bwd(B, Cin, C, N, _ptr(f), _ptr(w_f), _ptr(dg_cm), _ptr(df) if need_f else None, _ptr(dw_f),
                                  _ptr(work2), wb2, _stream(f)), "pointwise_conv_backward")
        dw = _join_columns(dw_dp, dw_f).view(ctx.wshape)
        return df, None, None, None, dw, dgamma, dbeta, None, None, None, None, None


@torch.no_grad()
def grouped_conv_bn_eval(f, dp, idx, weight, bn, relu):
    """GroupedConvBN in inference mode (running statistics), no gradient -> x1 (B,C,M,32)"""
    f, dp, idx = f.contiguous(), dp.contiguous(), idx.contiguous()
    B, Cin, N = f.shape
    _, M, K = idx.shape
    C = weight.shape[0]
    dev = f.device
    lib = _lib.load()
    w2 = weight.reshape(C, Cin + 3)
    w_dp, w_f = _split_columns(w2, 3)
    g_cm = torch.empty(B, C, N, dtype=torch.float32, device=dev)
    g_pm = torch.empty(B, N, C, dtype=torch.float32, device=dev)
    x1 = torch.empty(B, C, M, K, dtype=torch.float32, device=dev)
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    with torch.cuda.device(dev):
        _pw_forward(lib, False, B, Cin, C, N, f, w_f, None, g_cm)
        _lib.check(lib.amc3d_grouped_conv_bn_forward(
            B, C, N, M, K, 0, int(bool(relu)), float(bn.eps), 0.0, _ptr(g_cm), _ptr(idx), _ptr(dp), _ptr(w_dp), None,
            _ptr(bn.weight), _ptr(bn.bias), _ptr(g_pm), _ptr(x1), _ptr(bn.running_mean), _ptr(invstd), None, None, None,
            None, None, 0, None, None, 0, _stream(f)), "grouped_conv_bn_forward")
    return x1


def grouped_conv_bn_supported(cout, nsample):
    return bool(_lib.load().amc3d_grouped_conv_bn_supported(int(cout), int(nsample)))


@torch.no_grad()
def local_aggregation_eval(f, dp, idx, weight, bn, relu):
    """the same layer in inference mode (running statistics), no gradient"""
    f, dp, idx = f.contiguous(), dp.contiguous(), idx.contiguous()
    B, Cin, N = f.shape
    _, M, K = idx.shape
    C = weight.shape[0]
    dev = f.device
    lib = _lib.load()
    w2 = weight.reshape(C, Cin + 3)
    w_dp, w_f = _split_columns(w2, 3)
    g_cm = torch.empty(B, C, N, dtype=torch.float32, device=dev)
    g_pm = torch.empty(B, N, C, dtype=torch.float32, device=dev)
    pooled = torch.empty(B, C, M, dtype=torch.float32, device=dev)
    ystar = torch.empty_like(pooled)
    arg = torch.empty(B, C, M, dtype=torch.uint8, device=dev)
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    with torch.cuda.device(dev):
        _pw_forward(lib, False, B, Cin, C, N, f, w_f, None, g_cm)
        _lib.check(lib.amc3d_local_aggregation_forward(
            B, C, N, M, K, 0, int(bool(relu)), float(bn.eps), 0.0, _ptr(g_cm), _ptr(idx), _ptr(dp), _ptr(w_dp), None,
            _ptr(bn.weight), _ptr(bn.bias), _ptr(g_pm), _ptr(pooled), _ptr(arg), _ptr(ystar), _ptr(bn.running_mean),
            _ptr(invstd), None, None, None, None, None, 0, None, None, 0, _stream(f)), "local_aggregation_forward")
    return pooled


class PointwiseConv(Function):
    """y = conv1x1(x, weight, bias): nn.Conv1d / nn.Conv2d with kernel size 1 (models/layers/conv.py:8-21) on the
    fp32 MFMA kernels of csrc/pwconv.hip.  x (B,Cin,*spatial) fp32, weight (Cout,Cin,1[,1]), bias (Cout) or None."""

    @staticmethod
    def forward(ctx, x, weight, bias, bf16=False):
        _need_gpu(x, weight)
        _need_dtype(torch.float32, x=x, weight=weight, bias=bias)
        x = x.contiguous()
        B, Cin = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        Cout = weight.shape[0]
        assert weight.numel() == Cout * Cin, "pointwise_conv needs a 1x1 kernel"
        w2 = weight.reshape(Cout, Cin).contiguous()
        y = torch.empty((B, Cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device), timing.span("pointwise_conv_forward", 4 * B * P * (Cin + Cout),
                                                      2.0 * B * P * Cin * Cout):
            _pw_forward(lib, bf16, B, Cin, Cout, P, x, w2, bias.contiguous() if bias is not None else None, y)
        ctx.save_for_backward(x, w2)
        ctx.wshape = tuple(weight.shape)
        ctx.has_bias = bias is not None
        ctx.bf16 = bool(bf16)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        B, Cin = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        Cout = w2.shape[0]
        dy = dy.contiguous()
        dev = dy.device
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty(Cout, Cin, dtype=torch.float32, device=dev) if need_w else None
        _, wbytes, bwd = _pw(_lib.load(), ctx.bf16)
        wb = int(wbytes(B, Cin, Cout, P))  # weight-gradient partials and / or the split-K partials of dx
        work = torch.empty(max(wb, 4), dtype=torch.uint8, device=dev)
        flops = 2.0 * B * P * Cin * Cout * (int(need_x) + int(need_w))
        with torch.cuda.device(dev), timing.span("pointwise_conv_backward",
                                                 4 * B * P * (Cout + Cin * int(need_x) + Cin * int(need_w)), flops,
                                                 moved=4 * B * P * ((Cin + Cout) * int(need_x) + (Cin + Cout) * int(need_w))):
            _lib.check(bwd(B, Cin, Cout, P, _ptr(x), _ptr(w2), _ptr(dy), _ptr(dx) if need_x else None,
                           _ptr(dw) if need_w else None, _ptr(work), wb, _stream(dy)), "pointwise_conv_backward")
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Cout, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.load().amc3d_bias_grad(B, Cout, P, _ptr(dy), _ptr(db), _stream(dy)), "bias_grad")
        return dx, (dw.view(ctx.wshape) if need_w else None), db, None


def pointwise_conv(x, weight, bias=None, bf16=False):
    return PointwiseConv.apply(x, weight, bias, bf16)


def _split_columns(w2, c1):
    """w2 (rows, c1 + c2) fp32 on the GPU -> contiguous (rows, c1), (rows, c2) in one launch (no autograd)"""
    w2 = w2.detach().contiguous()
    rows, c2 = w2.shape[0], w2.shape[1] - c1
    a = torch.empty(rows, c1, dtype=torch.float32, device=w2.device)
    b = torch.empty(rows, c2, dtype=torch.float32, device=w2.device)
    with torch.cuda.device(w2.device):
        _lib.check(_lib.load().amc3d_split_columns(rows, c1, c2, _ptr(w2), _ptr(a), _ptr(b), _stream(w2)), "split_columns")
    return a, b


def _join_columns(a, b):
    """(rows, c1), (rows, c2) -> (rows, c1 + c2) in one launch"""
    a, b = a.contiguous(), b.contiguous()
    rows, c1, c2 = a.shape[0], a.shape[1], b.shape[1]
    w = torch.empty(rows, c1 + c2, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().amc3d_join_columns(rows, c1, c2, _ptr(a), _ptr(b), _ptr(w), _stream(a)), "join_columns")
    return w


class SplitWeight(Function):
    """(w[:, :c1], w[:, c1:]) of a 1x1-conv weight (Cout, C1+C2, 1) as two contiguous (Cout, C) matrices -- the two halves of a
    FeaturePropogation conv applied to the skip features and to the coarse features separately.  As torch slices the
    backward is zeros + copy per slice, an add, and zeros + copy for the select (7 launches per decoder level); here it is
    one concatenation."""

    @staticmethod
    def forward(ctx, weight, c1):
        w = weight.reshape(weight.shape[0], -1)
        ctx.wshape = tuple(weight.shape)
        if w.is_cuda and w.dtype == torch.float32:
            return _split_columns(w, c1)
        return w[:, :c1].contiguous(), w[:, c1:].contiguous()

    @staticmethod
    def backward(ctx, g1, g2):
        if g1.is_cuda and g1.dtype == torch.float32 and g2.dtype == torch.float32:
            return _join_columns(g1, g2).view(ctx.wshape), None
        return torch.cat((g1, g2), dim=1).view(ctx.wshape), None


def split_weight(weight, c1):
    return SplitWeight.apply(weight, c1)


class MaskedRefineDual(Function):
    """RefinementMethod.DualMasks with fusion 'MIN' (openpoints/AMContrast3D/MaskedRefine.py:55-131) on csrc/refine.hip: two
    launches forward, two backward, instead of ~25 tensor operations per decoder level.  feature (B,D,n) fp32, ambiguity
    (B,1,n) or (B*n,[1]) fp32 (no gradient: it only enters comparisons and an arg-min), neighbor_idx (B*n, K-1) int32 (a view
    with a row stride is fine).  -> (refined feature, count of refined points as a 0-dim int32 tensor)."""

    @staticmethod
    def forward(ctx, feature, ambiguity, neighbor_idx, threshold, threshold_max, gamma):
        _need_gpu(feature, ambiguity, neighbor_idx)
        _need_dtype(torch.float32, feature=feature, ambiguity=ambiguity)
        _need_dtype(torch.int32, neighbor_idx=neighbor_idx)
        f = feature.contiguous()
        a = ambiguity.detach().contiguous().view(-1)
        B, D, n = f.shape
        assert a.numel() == B * n and neighbor_idx.dim() == 2 and neighbor_idx.shape[0] == B * n
        nptr, k, stride, keep = _nbr_view(neighbor_idx)
        dev = f.device
        lib = _lib.load()
        out = torch.empty_like(f)
        best = torch.empty(B * n, dtype=torch.int32, device=dev)
        mask = torch.empty(B * n, dtype=torch.uint8, device=dev)
        count = torch.empty((), dtype=torch.int32, device=dev)
        work = torch.empty(max(int(lib.amc3d_masked_refine_workspace_ints(B * n)), 1), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev), timing.span("masked_refine_forward", f.numel() * 12 + B * n * (4 * k + 13)):
            _lib.check(lib.amc3d_masked_refine_forward(B, D, n, k, stride, _ptr(f), _ptr(a), nptr, float(threshold),
                                                       float(threshold_max), float(gamma), _ptr(out), _ptr(best), _ptr(mask),
                                                       _ptr(count), _ptr(work), _stream(f)), "masked_refine_forward")
        ctx.save_for_backward(best, mask)
        ctx.gamma = float(gamma)
        ctx.mark_non_differentiable(count)
        return out, count

    @staticmethod
    def backward(ctx, dout, _dcount):
        best, mask = ctx.saved_tensors
        dout = dout.contiguous()
        B, D, n = dout.shape
        df = torch.empty_like(dout)
        with torch.cuda.device(dout.device), timing.span("masked_refine_backward", dout.numel() * 12):
            _lib.check(_lib.load().amc3d_masked_refine_backward(B, D, n, ctx.gamma, _ptr(dout), _ptr(best), _ptr(mask), _ptr(df),
                                                                _stream(dout)), "masked_refine_backward")
        return df, None, None, None, None, None


class SAResidual(Function):
    """out = relu(y + skipconv(f[:, :, fps_idx])): the residual branch of a strided SetAbstraction block with use_res
    (pointnext_AA.py:157-168: torch.gather -> Conv1d with bias -> add -> ReLU) on csrc/sa_res.hip -- one launch forward;
    backward: ReLU mask + bias gradient + zero-fill, W^T . g scattered to the sampled columns, and the weight gradient.
    y (B,Cout,M) pooled main branch, f (B,Cin,N), fps_idx (B,M) int32, weight (Cout,Cin,1), bias (Cout) or None."""

    @staticmethod
    def forward(ctx, y, f, fps_idx, weight, bias, dup_flag=None):
        _need_gpu(y, f, fps_idx, weight)
        if dup_flag is not None:
            _need_gpu(dup_flag)
            _need_dtype(torch.int32, dup_flag=dup_flag)
        _need_dtype(torch.float32, y=y, f=f, weight=weight, bias=bias)
        _need_dtype(torch.int32, fps_idx=fps_idx)
        y, f, fps_idx = y.contiguous(), f.contiguous(), fps_idx.contiguous()
        B, Cin, N = f.shape
        Cout, M = weight.shape[0], fps_idx.shape[1]
        assert y.shape == (B, Cout, M) and fps_idx.shape[0] == B and weight.numel() == Cout * Cin
        w2 = weight.reshape(Cout, Cin).contiguous()
        bias = bias.contiguous() if bias is not None else None
        out = torch.empty_like(y)
        keep = any(ctx.needs_input_grad)
        fi = torch.empty(B, Cin, M, dtype=torch.float32, device=f.device) if keep else None
        with torch.cuda.device(f.device), timing.span("sa_residual_forward", 4 * B * M * (Cin * (2 if keep else 1) + 2 * Cout) + 4 * B * M,
                                                      2.0 * B * M * Cin * Cout):
            _lib.check(_lib.load().amc3d_sa_residual_forward(B, Cin, Cout, N, M, _ptr(f), _ptr(fps_idx), _ptr(w2),
                                                             _ptr(bias) if bias is not None else None, _ptr(y), _ptr(out),
                                                             _ptr(fi) if keep else None, _stream(f)), "sa_residual_forward")
        if keep:
            ctx.save_for_backward(out, fi, fps_idx, w2, *([dup_flag] if dup_flag is not None else []))
        ctx.n, ctx.wshape, ctx.has_bias = N, tuple(weight.shape), bias is not None
        ctx.mark_non_differentiable(fps_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, fi, fps_idx, w2 = ctx.saved_tensors[:4]
        dup_flag = ctx.saved_tensors[4] if len(ctx.saved_tensors) > 4 else None
        B, Cout, M = out.shape
        Cin, N = w2.shape[1], ctx.n
        dout = dout.contiguous()
        dev = dout.device
        need_f, need_w = ctx.needs_input_grad[1], ctx.needs_input_grad[3]
        need_b = ctx.has_bias and ctx.needs_input_grad[4]
        g = torch.empty_like(out)
        df = torch.empty(B, Cin, N, dtype=torch.float32, device=dev) if need_f else None
        dw = torch.empty(Cout, Cin, dtype=torch.float32, device=dev) if need_w else None
        db = torch.empty(Cout, dtype=torch.float32, device=dev) if need_b else None
        lib = _lib.load()
        wb = int(lib.amc3d_sa_residual_workspace_bytes(B, Cin, Cout, M))
        work = torch.empty(max(wb, 4), dtype=torch.uint8, device=dev)
        nbytes = 4 * B * M * 3 * Cout + (4 * B * Cin * N + 4 * B * M * Cout) * int(need_f) + 4 * B * M * (Cin + Cout) * int(need_w)
        with torch.cuda.device(dev), timing.span("sa_residual_backward", nbytes,
                                                 2.0 * B * M * Cin * Cout * (int(need_f) + int(need_w))):
            _lib.check(lib.amc3d_sa_residual_backward(B, Cin, Cout, N, M, _ptr(dout), _ptr(out), _ptr(fi), _ptr(fps_idx),
                                                      _ptr(dup_flag) if dup_flag is not None else None,
                                                      _ptr(w2), _ptr(g), _ptr(df) if need_f else None,
                                                      _ptr(dw) if need_w else None, _ptr(db) if need_b else None,
                                                      _ptr(work), wb, _stream(dout)), "sa_residual_backward")
        return g, df, None, (dw.view(ctx.wshape) if need_w else None), db, None


def sa_residual(y, f, fps_idx, weight, bias, dup_flag=None):
    """dup_flag: index_duplicates(fps_idx, N) of the plan, or None (the backward then adds with float atomics, right for any picks)"""
    return SAResidual.apply(y, f, fps_idx, weight, bias, dup_flag)


@torch.no_grad()
def index_duplicates(idx, n):
    """int32 [1] on the device: 1 if some row of idx (B, M) int32 into n points repeats an index, else 0 (no host sync)"""
    _need_gpu(idx)
    _need_dtype(torch.int32, idx=idx)
    idx = idx.contiguous()
    B, M = idx.shape
    flag = torch.empty(1, dtype=torch.int32, device=idx.device)
    lib = _lib.load()
    wb = int(lib.amc3d_index_duplicates_workspace_bytes(B, int(n)))
    work = torch.empty(max(wb, 4), dtype=torch.uint8, device=idx.device)
    with torch.cuda.device(idx.device):
        _lib.check(lib.amc3d_index_duplicates(B, int(n), M, _ptr(idx), _ptr(flag), _ptr(work), wb, _stream(idx)), "index_duplicates")
    return flag


def _library_wgrad(dy3, x3):
    """dW (Cout,Cin) = sum_b dy[b] . x[b]^T of a deep short layer.  Longer layers: batched library GEMM + sum over the
    batch.  The shortest ones (< 1024 positions per cloud, >= 128 channels: the two coarsest FeaturePropagation levels), where
    the batched form hits a slow library heuristic (256x256x375: 97 us) and one GEMM over (batch x positions) needs
    transposed copies of both operands: this library's streaming weight-gradient kernel (csrc/gemm.hip, fixed-order
    partial sums; the step takes the same time, measured, with 8 copies and 4 library launches fewer).  The form is a pure
    function of the shape -- the same in eager and captured runs and on every rank; AMC3D_WGRAD_FORM=bmm|flat|own
    overrides it."""
    import os
    B, Cout, P = dy3.shape
    Cin = x3.shape[1]
    form = os.environ.get("AMC3D_WGRAD_FORM") or ("own" if P < 1024 and min(Cin, Cout) >= 128 else "bmm")
    if form == "own":
        lib = _lib.load()
        dw = torch.empty(Cout, Cin, dtype=torch.float32, device=dy3.device)
        wb = int(lib.amc3d_pointwise_conv_workspace_bytes(B, Cin, Cout, P))
        work = torch.empty(max(wb, 4), dtype=torch.uint8, device=dy3.device)
        with torch.cuda.device(dy3.device):
            _lib.check(lib.amc3d_pointwise_conv_backward(B, Cin, Cout, P, _ptr(x3), None, _ptr(dy3), None, _ptr(dw), _ptr(work),
                                                         wb, _stream(dy3)), "pointwise_conv_backward")
        return dw
    if form == "flat":
        return torch.matmul(dy3.transpose(0, 1).reshape(Cout, B * P), x3.transpose(0, 1).reshape(Cin, B * P).t())
    return torch.bmm(dy3, x3.transpose(1, 2)).sum(0)


class LibraryGemmConv(Function):
    """1x1 convolution of a deep, short layer (>= 64 channels on both sides, < 65536 positions: SetAbstraction 4 and the
    coarse FeaturePropagation stages) as three plain library GEMMs.  These are small MFMA-bound GEMMs that rocBLAS does
    at ~80 TFLOP/s, twice what csrc/gemm.hip reaches at this size; what is avoided is the convolution library's
    weight-gradient path, which wraps an implicit-GEMM kernel in NCHW<->NHWC transposes (60-70 us per layer).
    x (B,Cin,*spatial) fp32 contiguous, weight (Cout,Cin,1[,1]), no bias."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        B, Cin = x.shape[0], x.shape[1]
        Cout = weight.shape[0]
        w2 = weight.reshape(Cout, Cin)
        timing.note("library_gemm_conv")
        import os
        # under bf16 autocast (use_amp, main_AA.py:389): bf16 operands for the three library GEMMs, fp32 accumulation inside the
        # library, fp32 tensors outside -- cfg 5 (XL + ++, 1 x 120000): 20.7 -> 19.7 ms per step (fp32: 20.2); AMC3D_LIB_FP32=1 keeps fp32
        ctx.bf16 = bool(torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
                        and not os.environ.get("AMC3D_LIB_FP32"))
        # bmm with the weight expanded along the batch (stride 0): torch.matmul(2-d, 3-d) would fold the batch into one
        # GEMM by way of a transposed copy of x.
        with torch.autocast("cuda", enabled=False):
            if ctx.bf16:  # bf16 operands (fp32 accumulation inside the library), fp32 tensors outside
                x16, w16 = x.view(B, Cin, -1).to(torch.bfloat16), w2.to(torch.bfloat16)
                y = torch.bmm(w16.unsqueeze(0).expand(B, Cout, Cin), x16).float()
                ctx.save_for_backward(x16, w16)
            else:
                y = torch.bmm(w2.unsqueeze(0).expand(B, Cout, Cin), x.view(B, Cin, -1))
                ctx.save_for_backward(x, w2)
        ctx.wshape = tuple(weight.shape)
        ctx.xshape = tuple(x.shape)
        return y.view((B, Cout) + tuple(x.shape[2:]))

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        B, Cin = x.shape[0], x.shape[1]
        dy3 = dy.contiguous().view(B, w2.shape[0], -1)
        if ctx.bf16:
            with torch.autocast("cuda", enabled=False):
                dy16 = dy3.to(torch.bfloat16)
                dx = (torch.bmm(w2.t().unsqueeze(0).expand(B, Cin, w2.shape[0]), dy16).float().view(ctx.xshape)
                      if ctx.needs_input_grad[0] else None)
                dw = (torch.bmm(dy16, x.transpose(1, 2)).float().sum(0).view(ctx.wshape) if ctx.needs_input_grad[1] else None)
            return dx, dw
        with torch.autocast("cuda", enabled=False):
            dx = (torch.bmm(w2.t().unsqueeze(0).expand(B, Cin, w2.shape[0]), dy3).view(x.shape)
                  if ctx.needs_input_grad[0] else None)
            dw = _library_wgrad(dy3, x.view(B, Cin, -1)).view(ctx.wshape) if ctx.needs_input_grad[1] else None
        return dx, dw


def library_gemm_conv(x, weight):
    return LibraryGemmConv.apply(x, weight)


def confusion_update(cm, invalid, logits, target, ignore_index=None):
    """cm (v,v) int64 += histogram of (target, argmax_c logits) for logits (B,C,N) fp32 and target (B,N) int64 -- the
    trainer's `cm.update(logits.argmax(dim=1), target)` (main_AA.py:414-415) as one launch; v = C (+1 with an ignore label:
    such points count in the extra row / column, openpoints/utils/metrics.py).  invalid (1) int64 += out-of-range targets."""
    _need_gpu(logits, target, cm, invalid)
    _need_dtype(torch.float32, logits=logits)
    _need_dtype(torch.int64, target=target, cm=cm, invalid=invalid)
    logits, target = logits.contiguous(), target.contiguous()
    B, C, N = logits.shape
    v = C + (1 if ignore_index is not None else 0)
    assert cm.is_contiguous() and cm.shape == (v, v) and target.shape == (B, N)
    with torch.cuda.device(logits.device), timing.span("confusion_update", logits.numel() * 4 + target.numel() * 8):
        _lib.check(_lib.load().amc3d_confusion_update(B, C, N, _ptr(logits), _ptr(target), int(ignore_index if ignore_index is not None else 0),
                                                      int(ignore_index is not None), _ptr(cm), _ptr(invalid), _stream(logits)),
                   "confusion_update")


class CrossEntropyMean(Function):
    """nn.CrossEntropyLoss()(logits.transpose(1, 2).reshape(-1, C), target.flatten()) -- default arguments: mean
    over the targets != ignore_index -- on the channel-major logits (B, C, N) as the model returns them
    (loss/build.py:328,338-340), in one pass forward and one backward."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        _need_gpu(logits, target)
        logits = logits.contiguous()
        B, C = logits.shape[0], logits.shape[1]
        N = logits[0, 0].numel()
        target = target.reshape(B, N).contiguous()
        assert target.dtype == torch.int64 and logits.dtype == torch.float32
        dev = logits.device
        lse = torch.empty(B, N, dtype=torch.float32, device=dev)
        mean_cnt = torch.empty(2, dtype=torch.float32, device=dev)
        lib = _lib.load()
        wb = int(lib.amc3d_cross_entropy_workspace_bytes(B, N))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev), timing.span("cross_entropy_forward", B * N * (4 * C + 12)):
            _lib.check(lib.amc3d_cross_entropy_forward(B, C, N, _ptr(logits), _ptr(target), int(ignore_index), _ptr(lse),
                                                       _ptr(mean_cnt), _ptr(work), wb, _stream(logits)), "cross_entropy_forward")
        ctx.save_for_backward(logits, target, lse, mean_cnt)
        ctx.ignore_index = int(ignore_index)
        return mean_cnt[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, lse, mean_cnt = ctx.saved_tensors
        B, C = logits.shape[0], logits.shape[1]
        N = logits[0, 0].numel()
        g = g.contiguous().to(torch.float32)
        d = torch.empty_like(logits)
        with torch.cuda.device(logits.device), timing.span("cross_entropy_backward", B * N * (8 * C + 12)):
            _lib.check(_lib.load().amc3d_cross_entropy_backward(B, C, N, _ptr(logits), _ptr(target), ctx.ignore_index,
                                                                _ptr(lse), _ptr(mean_cnt), _ptr(g), _ptr(d),
                                                                _stream(logits)), "cross_entropy_backward")
        return d, None, None


def cross_entropy_mean(logits, target, ignore_index=-100):
    return CrossEntropyMean.apply(logits, target, ignore_index)


class SATail(Function):
    """pooled (B,C2,M) = max_k [relu2](bn2(conv2(relu(bn1(y1)))))  -- the tail of a two-layer SetAbstraction block
    (pointnext_AA.py:104-127, 164-166) from the first conv's raw output y1 (B,C1,M,32), with batch statistics for both
    BatchNorms, without materialising any (B,C,M,32) tensor after y1 (csrc/sa_tail.hip re-creates them per tile).
    `bn1`, `bn2`: the nn.BatchNorm2d modules whose running buffers are updated (or None)."""

    @staticmethod
    def forward(ctx, y1, g1, b1, eps1, w2, g2, b2, eps2, relu2, bn1=None, bn2=None):
        _need_gpu(y1, g1, b1, w2, g2, b2)
        y1 = y1.contiguous()
        B, C1, M, K = y1.shape
        C2 = w2.shape[0]
        dev = y1.device
        lib = _lib.load()
        assert w2.numel() == C2 * C1 and lib.amc3d_sa_tail_supported(C1, C2, K)
        w2f = w2.reshape(C2, C1).contiguous()
        mean1 = torch.empty(C1, dtype=torch.float32, device=dev)
        invstd1, var1 = torch.empty_like(mean1), torch.empty_like(mean1)
        mean2 = torch.empty(C2, dtype=torch.float32, device=dev)
        invstd2, var2 = torch.empty_like(mean2), torch.empty_like(mean2)
        pooled = torch.empty(B, C2, M, dtype=torch.float32, device=dev)
        # the raw extreme the pool selected and the neighbour that held it: the backward then recomputes nothing (csrc/sa_tail.hip)
        zext = torch.empty(B, C2, M, dtype=torch.float32, device=dev)
        arg = torch.empty(B, C2, M, dtype=torch.uint8, device=dev)
        work1, wb1 = _bn_ws(C1, dev)
        wb = int(lib.amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        mom2, rm2, rv2, nbt2 = _bn_running_args(bn2)
        flops = 2.0 * B * M * K * C1 * C2 * 2  # the 1x1 conv is evaluated twice (statistics pass, max pass)
        with torch.cuda.device(dev), timing.span("sa_tail_forward", y1.numel() * 4 + pooled.numel() * 5, flops, moved=y1.numel() * 4 * 3 + pooled.numel() * 5):
            _lib.check(lib.amc3d_bn_stats(B, C1, M * K, float(eps1), _ptr(y1), _ptr(mean1), _ptr(invstd1), _ptr(var1),
                                          _ptr(work1), wb1, _stream(y1)), "bn_stats")
            _lib.check(lib.amc3d_sa_tail_forward(B, C1, C2, M, K, _ptr(y1), _ptr(mean1), _ptr(invstd1), _ptr(g1), _ptr(b1),
                                                 _ptr(w2f), _ptr(g2), _ptr(b2), float(eps2), mom2, int(bool(relu2)),
                                                 _ptr(pooled), _ptr(mean2), _ptr(invstd2), _ptr(var2), rm2, rv2,
                                                 nbt2, _ptr(zext), _ptr(arg), _ptr(work), wb, _stream(y1)), "sa_tail_forward")
        if bn1 is not None and bn1.track_running_stats and bn1.running_mean is not None:
            bn_update_running(bn1, mean1, var1)
        if bn2 is not None and bn2.track_running_stats and bn2.running_mean is not None and bn2.momentum is None:
            bn_update_running(bn2, mean2, var2)  # cumulative average: its own launch
        ctx.save_for_backward(y1, g1, b1, w2f, g2, b2, mean1, invstd1, mean2, invstd2, zext, arg)
        ctx.relu2, ctx.wshape = bool(relu2), tuple(w2.shape)
        ctx.pool_seq = _next_pool_seq() if _pool_log is not None else None
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        y1, g1, b1, w2f, g2, b2, mean1, invstd1, mean2, invstd2, zext, arg_ext = ctx.saved_tensors
        B, C1, M, K = y1.shape
        C2 = w2f.shape[0]
        dev = y1.device
        dpooled = dpooled.contiguous()
        lib = _lib.load()
        dx1 = torch.empty_like(y1)
        dw2 = torch.empty(C2, C1, dtype=torch.float32, device=dev)
        dg2, db2 = torch.empty_like(g2), torch.empty_like(b2)
        wb = int(lib.amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        dy1 = torch.empty_like(y1)
        dg1, db1 = torch.empty_like(g1), torch.empty_like(b1)
        work1, wb1 = _bn_ws(C1, dev, extra=C1 * 8)
        arg = None
        if ctx.pool_seq is not None and _pool_log is not None:  # tests: the forward pass keeps no arg-max, backward re-derives it
            arg = _pool_log[ctx.pool_seq] = torch.empty(B, C2, M, dtype=torch.uint8, device=dev)
        flops = 2.0 * B * M * K * C1 * C2 * 4  # recompute twice + dx1 + dW2
        with torch.cuda.device(dev), timing.span("sa_tail_backward", y1.numel() * 4 * 2 + dpooled.numel() * 10, flops, moved=y1.numel() * 4 * 6 + dpooled.numel() * 10):
            _lib.check(lib.amc3d_sa_tail_backward(B, C1, C2, M, K, _ptr(y1), _ptr(mean1), _ptr(invstd1), _ptr(g1), _ptr(b1),
                                                  _ptr(w2f), _ptr(mean2), _ptr(invstd2), _ptr(g2), _ptr(b2), int(ctx.relu2),
                                                  _ptr(dpooled), _ptr(zext), _ptr(arg_ext), _ptr(dx1), 0, _ptr(dw2), _ptr(dg2), _ptr(db2),
                                                  _ptr(arg) if arg is not None else None,
                                                  _ptr(work), wb, _stream(y1)), "sa_tail_backward")
            # BN1 + ReLU backward on the raw y1 (csrc/bn.hip)
            _lib.check(lib.amc3d_bn_backward(B, C1, M * K, 1, 1, _ptr(y1), _ptr(dx1), None, _ptr(mean1), _ptr(invstd1),
                                             _ptr(g1), _ptr(b1), _ptr(dy1), _ptr(dg1), _ptr(db1), _ptr(work1), wb1,
                                             _stream(y1)), "bn_backward")
        return dy1, dg1, db1, None, dw2.view(ctx.wshape), dg2, db2, None, None, None, None


_identity_bn = {}  # (C, device) -> (zeros, ones): BatchNorm parameters under which relu(bn(x)) == x for x >= 0


class SATailActivated(Function):
    """pooled (B,C2,M) = max_k [relu2](bn2(conv2(x1))) from the ACTIVATED first-layer output x1 = relu(bn1(y1)) >= 0
    (ops.GroupedConvBN): SATail's recomputing kernels with an identity first BatchNorm; the gradient it returns is dx1."""

    @staticmethod
    def forward(ctx, x1, w2, g2, b2, eps2, relu2, bn2=None):
        _need_gpu(x1, w2, g2, b2)
        x1 = x1.contiguous()
        B, C1, M, K = x1.shape
        C2 = w2.shape[0]
        dev = x1.device
        lib = _lib.load()
        assert w2.numel() == C2 * C1 and lib.amc3d_sa_tail_supported(C1, C2, K)
        key = (C1, str(dev))
        if key not in _identity_bn:
            _identity_bn[key] = (torch.zeros(C1, dtype=torch.float32, device=dev), torch.ones(C1, dtype=torch.float32, device=dev))
        zeros, ones = _identity_bn[key]
        w2f = w2.reshape(C2, C1).contiguous()
        mean2 = torch.empty(C2, dtype=torch.float32, device=dev)
        invstd2, var2 = torch.empty_like(mean2), torch.empty_like(mean2)
        pooled = torch.empty(B, C2, M, dtype=torch.float32, device=dev)
        zext = torch.empty(B, C2, M, dtype=torch.float32, device=dev)
        arg = torch.empty(B, C2, M, dtype=torch.uint8, device=dev)
        wb = int(lib.amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        mom2, rm2, rv2, nbt2 = _bn_running_args(bn2)
        with torch.cuda.device(dev), timing.span("sa_tail_forward", x1.numel() * 4 + pooled.numel() * 10, 2.0 * B * M * K * C1 * C2):
            _lib.check(lib.amc3d_sa_tail_forward(B, C1, C2, M, K, _ptr(x1), _ptr(zeros), _ptr(ones), _ptr(ones), _ptr(zeros),
                                                 _ptr(w2f), _ptr(g2), _ptr(b2), float(eps2), mom2, int(bool(relu2)),
                                                 _ptr(pooled), _ptr(mean2), _ptr(invstd2), _ptr(var2), rm2, rv2,
                                                 nbt2, _ptr(zext), _ptr(arg), _ptr(work), wb, _stream(x1)), "sa_tail_forward")
        if bn2 is not None and bn2.track_running_stats and bn2.running_mean is not None and bn2.momentum is None:
            bn_update_running(bn2, mean2, var2)
        ctx.save_for_backward(x1, w2f, g2, b2, mean2, invstd2, zeros, ones, zext, arg)
        ctx.relu2, ctx.wshape = bool(relu2), tuple(w2.shape)
        ctx.pool_seq = _next_pool_seq() if _pool_log is not None else None
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x1, w2f, g2, b2, mean2, invstd2, zeros, ones, zext, arg_ext = ctx.saved_tensors
        B, C1, M, K = x1.shape
        C2 = w2f.shape[0]
        dev = x1.device
        dpooled = dpooled.contiguous()
        lib = _lib.load()
        # x1 comes from GroupedConvBN, whose backward gathers position-major rows: the gradient is written as
        # (B,M,K,C1) and returned as a (B,C1,M,K) view of it -- no transpose pass between the two (393 MB at SA1)
        import os
        pm = C1 % 4 == 0 and C1 > 1 and not os.environ.get("AMC3D_SAT_DX1_CM")
        if pm:
            dx1_buf = torch.empty(B, M, K, C1, dtype=torch.float32, device=dev)
            dx1 = dx1_buf.permute(0, 3, 1, 2)
        else:
            dx1_buf = dx1 = torch.empty_like(x1)
        dw2 = torch.empty(C2, C1, dtype=torch.float32, device=dev)
        dg2, db2 = torch.empty_like(g2), torch.empty_like(b2)
        wb = int(lib.amc3d_sa_tail_workspace_bytes(B, C1, C2, M))
        work = torch.empty(max(wb, 8), dtype=torch.uint8, device=dev)
        arg = None
        if ctx.pool_seq is not None and _pool_log is not None:
            arg = _pool_log[ctx.pool_seq] = torch.empty(B, C2, M, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev), timing.span("sa_tail_backward", x1.numel() * 4 * 2 + dpooled.numel() * 10,
                                                 2.0 * B * M * K * C1 * C2 * 4, moved=x1.numel() * 4 * 3 + dpooled.numel() * 10):
            _lib.check(lib.amc3d_sa_tail_backward(B, C1, C2, M, K, _ptr(x1), _ptr(zeros), _ptr(ones), _ptr(ones), _ptr(zeros),
                                                  _ptr(w2f), _ptr(mean2), _ptr(invstd2), _ptr(g2), _ptr(b2), int(ctx.relu2),
                                                  _ptr(dpooled), _ptr(zext), _ptr(arg_ext), _ptr(dx1_buf), int(pm), _ptr(dw2), _ptr(dg2), _ptr(db2),
                                                  _ptr(arg) if arg is not None else None,
                                                  _ptr(work), wb, _stream(x1)), "sa_tail_backward")
        return dx1, dw2.view(ctx.wshape), dg2, db2, None, None, None


def sa_tail_supported(c1, c2, k):
    return bool(_lib.load().amc3d_sa_tail_supported(int(c1), int(c2), int(k)))


def sa_tail_pays(c1, c2):
    return bool(_lib.load().amc3d_sa_tail_pays(int(c1), int(c2)))
