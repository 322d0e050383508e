"""float64 NumPy restatement of OthelloNN inference -- TEST INFRASTRUCTURE ONLY.

Follows Net/OthelloNN.py:42-56 as driven by NNetWrapper.predict (Net/NNet.py:70-87):
input (B,n,n,2) NHWC {0,1}; 4 x [Conv2D 3x3 (same, same, valid, valid) -> BatchNormalization(axis=3,
epsilon=1e-3, moving statistics) -> ReLU]; Flatten in (h,w,c) order; 2 x [Dense -> BN -> ReLU]
(Dropout is identity at inference); softmax policy head reshaped (n,n); tanh value head.

Parity status: UNPINNED by reference vectors (TensorFlow 2.3.1 / Keras 2.4.3 are a third-party
dependency absent from /root/reference and from this image; the reference has no tests).  This is
the documented Keras arithmetic evaluated in float64; tests cross-check it against torch-CPU and
against oz_oracle_nn.c.

`weights` = list of 40 arrays in keras Model.get_weights() order (see oz_oracle_nn.c).
"""
import numpy as np

BN_EPS = 1e-3


def planes(own, opp, n, dtype=np.float64):
    own = np.asarray(own, dtype=np.uint64).ravel()
    opp = np.asarray(opp, dtype=np.uint64).ravel()
    x = np.zeros((own.size, n, n, 2), dtype=dtype)
    for r in range(n):
        for c in range(n):
            s = np.uint64(r * 8 + c)
            x[:, r, c, 0] = (own >> s) & np.uint64(1)
            x[:, r, c, 1] = (opp >> s) & np.uint64(1)
    return x


def _bn(x, g, b, mu, var):
    return (x - mu) / np.sqrt(var + BN_EPS) * g + b


def _conv3x3(x, k, bias, same):
    B, H, W, _ = x.shape
    if same:
        x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        Ho, Wo = H, W
    else:
        Ho, Wo = H - 2, W - 2
    out = np.zeros((B * Ho * Wo, k.shape[3]), dtype=x.dtype)
    for ky in range(3):
        for kx in range(3):
            # one (pixels x Cin) @ (Cin x Cout) product per tap on a contiguous copy of the shifted window (so that NumPy
            # hands it to BLAS: the strided 4-D form of the same product runs ~20x slower), taps summed in (ky, kx) order
            win = np.ascontiguousarray(x[:, ky:ky + Ho, kx:kx + Wo, :]).reshape(B * Ho * Wo, -1)
            out += win @ k[ky, kx]
    return out.reshape(B, Ho, Wo, -1) + bias


def forward_chunked(weights, own, opp, n, chunk=512, dtype=np.float64):
    """forward() over a large batch in slices of `chunk` positions (bounded memory: one slice's activations at a time)"""
    own = np.asarray(own, dtype=np.uint64).ravel()
    opp = np.asarray(opp, dtype=np.uint64).ravel()
    w = [np.asarray(a, dtype=dtype) for a in weights]
    pis, vs = [], []
    for s in range(0, own.size, chunk):
        pi, v = forward(w, own[s:s + chunk], opp[s:s + chunk], n, dtype)
        pis.append(pi); vs.append(v)
    return np.concatenate(pis), np.concatenate(vs)


def forward(weights, own, opp, n, dtype=np.float64, return_activations=False):
    w = [np.asarray(a, dtype=dtype) for a in weights]
    x = planes(own, opp, n, dtype)
    if w[0].shape[2] == 1:                         # BaseNN (Net/BaseNN.py:41-44): one plane, +1 mover / -1 opponent
        x = x[..., 0:1] - x[..., 1:2]
    acts = []
    for layer, same in enumerate((True, True, False, False)):
        k, bias, g, b, mu, var = w[6 * layer:6 * layer + 6]
        x = np.maximum(_bn(_conv3x3(x, k, bias, same), g, b, mu, var), 0)
        acts.append(x)
    x = x.reshape(x.shape[0], -1)                 # Flatten of NHWC -> (h, w, c) order
    for layer in (4, 5):
        k, bias, g, b, mu, var = w[6 * layer:6 * layer + 6]
        x = np.maximum(_bn(x @ k + bias, g, b, mu, var), 0)
        acts.append(x)
    logits = x @ w[36] + w[37]
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    pi = e / e.sum(axis=1, keepdims=True)
    v = np.tanh(x @ w[38] + w[39])[:, 0]
    if return_activations:
        return pi, v, acts, logits
    return pi, v
