"""OthelloNN parameter list in keras Model.get_weights() order (Net/OthelloNN.py:42-56).

index  array
0-5    conv1 kernel (3,3,2,C), bias (C), BN gamma, beta, moving_mean, moving_variance
6-11   conv2 kernel (3,3,C,C) ...      12-17 conv3 ...      18-23 conv4 ...
24-29  dense1 kernel ((n-4)^2*C, 1024), bias, BN x4
30-35  dense2 kernel (1024, 512), bias, BN x4
36-37  pi kernel (512, n*n), bias      38-39  v kernel (512, 1), bias
"""
import numpy as np


def onn_shapes(n, channels=512, in_channels=2):
    """in_channels = 2: OthelloNN (Net/OthelloNN.py), 1: BaseNN (Net/BaseNN.py) -- same trunk, one input plane"""
    C = channels
    shapes = []
    for cin in (in_channels, C, C, C):
        shapes += [(3, 3, cin, C), (C,), (C,), (C,), (C,), (C,)]
    shapes += [((n - 4) * (n - 4) * C, 1024)] + [(1024,)] * 5
    shapes += [(1024, 512)] + [(512,)] * 5
    shapes += [(512, n * n), (n * n,), (512, 1), (1,)]
    return shapes


def _glorot_uniform(rs, shape):
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        fan_in, fan_out = shape[2] * rf, shape[3] * rf
    else:
        fan_in, fan_out = shape
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)


def init_weights(n, seed=0, channels=512, randomize_all=False, in_channels=2):
    """Keras defaults (glorot_uniform kernels, zero bias, BN gamma=1 beta=0 mean=0 var=1).
    randomize_all=True also draws biases and BN statistics (numerics tests)."""
    rs = np.random.RandomState(seed)
    out = []
    for i, shp in enumerate(onn_shapes(n, channels, in_channels)):
        if len(shp) >= 2:
            out.append(_glorot_uniform(rs, shp))
            continue
        role = i % 6 if i < 36 else (1 if i in (37, 39) else 0)
        if not randomize_all:
            out.append(np.ones(shp, np.float32) if (i < 36 and role in (2, 5)) else np.zeros(shp, np.float32))
        elif i >= 36 or role == 1:
            out.append(rs.uniform(-0.1, 0.1, size=shp).astype(np.float32))           # biases
        elif role == 2:
            out.append(rs.uniform(0.5, 1.5, size=shp).astype(np.float32))            # gamma
        elif role == 3:
            out.append(rs.uniform(-0.2, 0.2, size=shp).astype(np.float32))           # beta
        elif role == 4:
            out.append(rs.uniform(-0.2, 0.2, size=shp).astype(np.float32))           # moving mean
        else:
            out.append(rs.uniform(0.5, 2.0, size=shp).astype(np.float32))            # moving variance
    return out
