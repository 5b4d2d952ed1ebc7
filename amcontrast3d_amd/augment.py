"""The loader's training transforms on the device, for a whole batch of cropped clouds at once.

The reference applies `cfgs/s3dis/default.yaml:33-43`'s chain -- ChromaticAutoContrast, PointsToTensor, PointCloudScaling,
PointCloudXYZAlign, PointCloudRotation, PointCloudJitter, ChromaticDropGPU, ChromaticNormalize (openpoints/transforms/
point_transform_cpu.py:8-19,192-209, point_transformer_gpu.py:70-89,135-164,216-229,267-311,373-409) -- per cloud inside
`S3DIS.__getitem__` (dataset/s3dis/s3dis.py:136-143), in loader worker processes.  With the step at 7 ms the six workers of
the reference's loader cannot keep up; `S3DISTrainAugment` does the same arithmetic for the (B,N,3) batch in two launches
(csrc/augment.hip), after `input_pipeline.crop_pc`.  Same parameters (the config's `kwargs`), same order of operations, the
random numbers drawn per cloud from a device generator (or passed in: `draws`); `heights` is, as in the reference, the gravity
coordinate of the cloud BEFORE the transforms.  Oracle: oracle/augment_ref.py, pinned by tests/golden/augment_s3dis.npz
(recorded from the reference's own classes); tests/test_gpu_augment.py."""
import ctypes

import torch

from . import _lib


class S3DISTrainAugment:
    def __init__(self, scale=(0.9, 1.1), angle=(0, 0, 1), jitter_sigma=0.005, jitter_clip=0.02, color_drop=0.2, contrast_p=0.2,
                 blend_factor=None, gravity_dim=2, color_mean=(0.5136457, 0.49523646, 0.44921124),
                 color_std=(0.18308958, 0.18415008, 0.19252081), **kwargs):
        self.scale, self.angle = (float(scale[0]), float(scale[1])), tuple(float(a) for a in angle)
        self.sigma, self.clip, self.color_drop, self.contrast_p = float(jitter_sigma), float(jitter_clip), float(color_drop), float(contrast_p)
        self.blend_factor, self.g = blend_factor, int(gravity_dim)
        self.color_mean, self.color_std = tuple(color_mean), tuple(color_std)
        self._const = {}

    def draw(self, B, N, device, generator=None):
        """the random numbers of one batch, per cloud, in the order the reference's classes draw them"""
        r = lambda *s: torch.rand(*s, device=device, generator=generator)  # noqa: E731
        import math
        bound = torch.tensor([a * math.pi for a in self.angle], device=device, dtype=torch.float64)
        return {"contrast": r(B) < self.contrast_p,
                "blend": r(B) if self.blend_factor is None else torch.full((B,), float(self.blend_factor), device=device),
                "scale_u": r(B, 3), "theta": (r(B, 3).double() * 2 - 1) * bound,
                "noise": torch.randn(B, N, 3, device=device, generator=generator), "drop": r(B) < self.color_drop}

    def params(self, d):
        """(B,24) per-cloud records of amc3d_augment_clouds from a set of draws"""
        B, dev = d["scale_u"].shape[0], d["scale_u"].device
        lo, hi = torch.tensor(self.scale[0], dtype=torch.float32), torch.tensor(self.scale[1], dtype=torch.float32)
        scale = d["scale_u"].float() * (hi - lo).to(dev) + lo.to(dev)  # point_transformer_gpu.py:151-152
        t = d["theta"].double()
        c, s, one, zero = torch.cos(t), torch.sin(t), torch.ones(B, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev)
        rx = torch.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], 1).view(B, 3, 3)
        ry = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], 1).view(B, 3, 3)
        rz = torch.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], 1).view(B, 3, 3)
        rot = (rx @ ry @ rz).float()  # expm of the axis generators (:272-274); the reference shuffles the order of the three
        p = torch.zeros(B, 24, dtype=torch.float32, device=dev)
        p[:, 0] = d["contrast"].float()
        p[:, 1] = d["blend"].float()
        p[:, 2:5] = scale
        p[:, 5:14] = rot.reshape(B, 9)
        p[:, 14] = d["drop"].float()
        return p

    def __call__(self, pos, color, draws=None, generator=None):
        """pos (B,N,3) fp32 cropped clouds, color (B,N,3) fp32 -> (pos', x (B,N,3) normalised colours, heights (B,N,1))"""
        if not (pos.is_cuda and color.is_cuda):
            raise RuntimeError("S3DISTrainAugment runs on the GPU only (there is no CPU path)")
        assert pos.dtype == color.dtype == torch.float32 and pos.shape == color.shape and pos.dim() == 3 and pos.shape[2] == 3
        pos, color = pos.contiguous(), color.contiguous()
        B, N, dev = pos.shape[0], pos.shape[1], pos.device
        d = draws if draws is not None else self.draw(B, N, dev, generator)
        par = self.params(d).contiguous()
        noise = d["noise"].to(torch.float32).contiguous()
        key = str(dev)
        if key not in self._const:
            self._const[key] = (torch.tensor(self.color_mean, dtype=torch.float32, device=dev),
                                torch.tensor(self.color_std, dtype=torch.float32, device=dev))
        mean, std = self._const[key]
        pos_out, x_out = torch.empty_like(pos), torch.empty_like(color)
        heights = torch.empty(B, N, 1, dtype=torch.float32, device=dev)
        lib = _lib.load()
        wb = int(lib.amc3d_augment_workspace_bytes(B))
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(dev):
            _lib.check(lib.amc3d_augment_clouds(B, N, self.g, self.sigma, self.clip, P(pos), P(color), P(noise), P(par), P(mean), P(std),
                                                P(pos_out), P(x_out), P(heights), P(work), wb,
                                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "augment_clouds")
        return pos_out, x_out, heights
