"""ORACLE -- test infrastructure only.  numpy restatement of the S3DIS training transform chain
(cfgs/s3dis/default.yaml:33-43) with every random draw passed in, each step citing the reference lines it follows
(paths relative to /root/reference/openpoints/transforms).  Pinned by tests/golden/augment_s3dis.npz, which
oracle/gen_golden.py records from the reference's own classes (tests/test_oracle_augment.py)."""
import numpy as np
from scipy.linalg import expm, norm


def chromatic_auto_contrast(x, take, blend):
    """point_transform_cpu.py:197-204 (x (n,3) float32 colours 0..255; blend: the Python float np.random.rand() returned)"""
    if not take:
        return x
    lo = np.min(x, 0, keepdims=True)
    hi = np.max(x, 0, keepdims=True)
    scale = 255 / (hi - lo)
    contrast = (x - lo) * scale
    return ((1 - float(blend)) * x + float(blend) * contrast).astype(np.float32)


def scaling(pos, scale_u, scale_min, scale_max):
    """point_transformer_gpu.py:149-160 without mirroring: scale = u * (max - min) + min per axis, pos *= scale"""
    scale = scale_u.astype(np.float32) * (np.float32(scale_max) - np.float32(scale_min)) + np.float32(scale_min)
    return pos * scale, scale


def xyz_align(pos, gravity_dim=2):
    """:82-85: pos -= mean over the points; pos[:, g] -= min"""
    # (the mean accumulated in double and rounded once: torch's and numpy's fp32 reductions differ from it, and from each other,
    # by their summation trees -- 3e-6 on a 24000-point cloud -- which is below what the comparison of the chain needs)
    pos = pos - np.mean(pos, axis=0, keepdims=True, dtype=np.float64).astype(np.float32)
    pos[:, gravity_dim] -= np.min(pos[:, gravity_dim])
    return pos


def rotation_matrix(theta, order=(0, 1, 2)):
    """:272-293: M(axis, theta) = expm(cross(eye(3), axis / |axis| * theta)) per axis, multiplied in a shuffled order
    (`order`; with angle = [0, 0, 1] two of the three are the identity and the order is immaterial), cast to float32"""
    mats = []
    for a in range(3):
        axis = np.zeros(3)
        axis[a] = 1
        mats.append(expm(np.cross(np.eye(3), axis / norm(axis) * float(theta[a]))))
    mats = [mats[i] for i in order]
    return (mats[0] @ mats[1] @ mats[2]).astype(np.float32)


def jitter(pos, noise, sigma, clip):
    """:222-225: pos += clamp(randn * sigma, -clip, clip)"""
    return pos + np.clip(noise.astype(np.float32) * np.float32(sigma), -np.float32(clip), np.float32(clip))


def chromatic_drop(x, take):
    """:378-381"""
    return np.zeros_like(x) if take else x


def chromatic_normalize(x, mean=(0.5136457, 0.49523646, 0.44921124), std=(0.18308958, 0.18415008, 0.19252081)):
    """:404-409: colours above 1 are taken as 0..255 and divided by 255; then (x - mean) / std in float32"""
    if x.max() > 1:
        x = x / np.float32(255.)
    return ((x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)).astype(np.float32)


def s3dis_train(coord, feat, d, scale=(0.9, 1.1), gravity_dim=2, jitter_sigma=0.005, jitter_clip=0.02):
    """The chain in the order of the config, as S3DIS.__getitem__ applies it (dataset/s3dis/s3dis.py:136-143).
    d: the draws {'contrast': bool, 'blend', 'scale_u' (3), 'theta' (3), 'noise' (n,3), 'drop': bool}.
    -> pos (n,3), x (n,3), heights (n,1) -- `heights` is the gravity coordinate of the UNtransformed cloud (s3dis.py:141-142 reads
    the numpy array it handed to the transforms; PointsToTensor copies -- np.array(...) -- before torch.from_numpy, so the
    in-place scaling / alignment never reaches it)."""
    x = chromatic_auto_contrast(feat.astype(np.float32), d["contrast"], d.get("blend", 0.0))
    pos, _ = scaling(coord.astype(np.float32), d["scale_u"], scale[0], scale[1])
    pos = xyz_align(pos, gravity_dim)
    heights = coord[:, gravity_dim:gravity_dim + 1].astype(np.float32)
    pos = pos @ rotation_matrix(d["theta"]).T
    pos = jitter(pos, d["noise"], jitter_sigma, jitter_clip)
    x = chromatic_normalize(chromatic_drop(x, d["drop"]))
    return pos.astype(np.float32), x, heights.astype(np.float32)
