"""Scalar helpers of the ambiguity estimation function (AMContrast3D/AEF/function.py:7-39)."""
import torch

_inf = 1e9
_eps = 1e-12


def inverse_sigmoid_function(cc, t, b):
    """a = 1 / (1 + t ** (b * cc)); the caller passes t = e (function.py:10-14)."""
    return 1 / (1 + t.pow(b * cc))


def square_distance(src, dst):
    """(B,N,C), (B,M,C) -> (B,N,M) squared distances in the expanded form
    -2 src.dst + |src|^2 + |dst|^2 (function.py:18-39)."""
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist
