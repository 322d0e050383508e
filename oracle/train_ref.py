"""float64 torch-CPU restatement of the reference's training step -- TEST INFRASTRUCTURE ONLY.

What it restates: `NNetWrapper.train` (Net/NNet.py:53-68) = `keras.Model.fit(x=boards, y=[pis, vs], batch_size, epochs)` on
the graph of Net/OthelloNN.py:42-56 / Net/BaseNN.py:41-57, compiled with
loss=['categorical_crossentropy', 'mean_squared_error'] and Adam(lr, clipvalue=0.5) (BaseNN: no clipvalue).

Parity status: UNPINNED by reference vectors -- TensorFlow 2.3.1 / Keras 2.4 are absent from /root/reference and from this
image and the reference has no tests, so nothing the reference computed can be replayed.  This file restates the
documented Keras / TF arithmetic; gradients come from torch autograd (float64), which is what the GPU kernels'
hand-derived backward pass is checked against.  Choices that follow TF 2.3.1's behaviour and are easy to get wrong:

* BatchNormalization in training mode normalises with the batch mean and the BIASED batch variance, epsilon 1e-3,
  momentum 0.99.  The 4-D conv BNs (axis=3) run Keras' fused kernel, whose moving-variance update uses the UNBIASED
  variance (x M/(M-1)); the 2-D dense BNs (axis=1) are not fused and feed the biased variance.
* The policy output goes through Reshape((n, n)) ('pi-reshaped') before the loss, so Keras no longer sees a Softmax op and
  takes the probability path of `categorical_crossentropy` with axis=-1 ON THE (B, n, n) TENSOR: each board ROW is
  renormalised (p / sum over the row), clipped to [1e-7, 1 - 1e-7], and -sum(t * log p) is taken per row; the loss is the mean
  over batch x rows.  (A reference quirk, reproduced: with one-hot targets only the row holding the 1 contributes.)
* v: mean squared error between (B, 1) predictions and (B,) targets expanded to (B, 1); total loss = pi loss + v loss.
* Dropout(rate) on the two dense blocks: inverted dropout.  Keras' random stream cannot be matched; masks come from the
  library's counter-based hash (`dropout_keep`), identical here and in the HIP kernel.
* Adam as tf.keras implements it: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); m, v updated; var -= lr_t * m / (sqrt(v) + 1e-7);
  `clipvalue` clips every gradient element to [-c, c] first.

Weights are the 40 arrays of keras get_weights() (see othellozero_amd/weights.py); indices 4,5 of every 6-block are the
non-trainable moving statistics.
"""
import numpy as np
import torch

BN_EPS = 1e-3
TRAINABLE = [i for i in range(40) if i >= 36 or i % 6 not in (4, 5)]

_M64 = (1 << 64) - 1


def _sm64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def dropout_keep(seed, step, layer, count, rate):
    """keep mask (bool[count]) of element idx of dropout layer `layer` (0, 1) at optimiser step `step` (0-based):
    u = top 24 bits of sm64(sm64(seed + K1*step) ^ K2*(layer+1) ^ K3*idx) / 2^24; keep iff u >= rate (float32 compare)."""
    with np.errstate(over="ignore"):
        a = _sm64(np.uint64(seed) + np.uint64(0x632BE59BD9B4E019) * np.uint64(step))
        idx = np.arange(count, dtype=np.uint64)
        h = _sm64(a ^ (np.uint64(layer + 1) * np.uint64(0xD1B54A32D192ED03)) ^ (idx * np.uint64(0x9E3779B97F4A7C15)))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(rate)


def planes(own, opp, n, in_channels=2):
    own = np.asarray(own, dtype=np.uint64).ravel()
    opp = np.asarray(opp, dtype=np.uint64).ravel()
    x = np.zeros((own.size, n, n, 2), dtype=np.float64)
    for r in range(n):
        for c in range(n):
            s = np.uint64(r * 8 + c)
            x[:, r, c, 0] = (own >> s) & np.uint64(1)
            x[:, r, c, 1] = (opp >> s) & np.uint64(1)
    if in_channels == 1:
        x = x[..., 0:1] - x[..., 1:2]
    return x


class TrainRef:
    def __init__(self, weights, n, lr=1e-3, clipvalue=0.5, dropout=0.3, momentum=0.99, seed=0):
        self.n, self.lr, self.clip, self.rate, self.mom, self.seed = n, lr, clipvalue, dropout, momentum, seed
        self.w = [torch.tensor(np.asarray(a, dtype=np.float64)) for a in weights]
        self.in_channels = self.w[0].shape[2]
        self.m = {i: torch.zeros_like(self.w[i]) for i in TRAINABLE}
        self.v = {i: torch.zeros_like(self.w[i]) for i in TRAINABLE}
        self.step = 0
        self.grads = None

    # ---- forward in training mode; returns losses and keeps the graph
    def _bn_train(self, z, blk, fused):
        g, b = self.w[blk + 2], self.w[blk + 3]
        dims = tuple(range(z.dim() - 1))
        M = z.numel() // z.shape[-1]
        mean = z.mean(dim=dims)
        var = ((z - mean) ** 2).mean(dim=dims)
        y = (z - mean) / torch.sqrt(var + BN_EPS) * g + b
        with torch.no_grad():
            upd_var = var * (M / max(M - 1, 1)) if fused else var
            self._new_stats[blk + 4] = self.w[blk + 4] * self.mom + mean.detach() * (1 - self.mom)
            self._new_stats[blk + 5] = self.w[blk + 5] * self.mom + upd_var.detach() * (1 - self.mom)
        return y

    def _relu(self, y, layer):
        """relu(y); where |y| < KINK the derivative is taken from `relu_masks[layer]` (the fp32 kernels' own decision)
        instead of this float64 sign: relu' is ambiguous there at fp32 precision, and with ~1e6 units per batch some unit
        always sits that close to the kink.  The forward value changes by < KINK."""
        r = torch.relu(y)
        if self.relu_masks is not None:
            m = torch.tensor(np.asarray(self.relu_masks[layer], dtype=np.float64).reshape(tuple(y.shape)))
            near = y.detach().abs() < 1e-5
            self.kink_units += int(near.sum())
            r = torch.where(near, y * m, r)
        return r

    def forward_backward(self, own, opp, pi_target, z_target, relu_masks=None):
        """relu_masks: optional list of 6 {0,1} arrays (conv blocks NHWC, dense blocks (B, units)), see _relu"""
        n = self.n
        self.relu_masks, self.kink_units = relu_masks, 0
        x = torch.tensor(planes(own, opp, n, self.in_channels))
        B = x.shape[0]
        for i in TRAINABLE:
            self.w[i].requires_grad_(True)
            self.w[i].grad = None
        self._new_stats = {}
        h = x.permute(0, 3, 1, 2)                                    # NCHW for torch conv
        for layer, same in enumerate((True, True, False, False)):
            blk = 6 * layer
            k = self.w[blk].permute(3, 2, 0, 1)                      # (3,3,Cin,Cout) -> (Cout,Cin,3,3)
            zc = torch.nn.functional.conv2d(h, k, self.w[blk + 1], padding=1 if same else 0)
            zc = self._bn_train(zc.permute(0, 2, 3, 1), blk, fused=True)
            h = self._relu(zc, layer).permute(0, 3, 1, 2)
        f = h.permute(0, 2, 3, 1).reshape(B, -1)                      # Flatten of NHWC
        for j, blk in enumerate((24, 30)):
            zd = f @ self.w[blk] + self.w[blk + 1]
            a = self._relu(self._bn_train(zd, blk, fused=False), 4 + j)
            if self.rate > 0:
                keep = torch.tensor(dropout_keep(self.seed, self.step, j, a.numel(), self.rate).reshape(a.shape))
                a = a * keep / (1.0 - self.rate)
            f = a
        logits = f @ self.w[36] + self.w[37]
        p = torch.softmax(logits, dim=1).reshape(B, n, n)
        v = torch.tanh(f @ self.w[38] + self.w[39])                  # (B, 1)
        t = torch.tensor(np.asarray(pi_target, dtype=np.float64).reshape(B, n, n))
        q = p / p.sum(dim=2, keepdim=True)
        q = torch.clamp(q, 1e-7, 1 - 1e-7)
        loss_pi = (-(t * torch.log(q)).sum(dim=2)).mean()
        zt = torch.tensor(np.asarray(z_target, dtype=np.float64).reshape(B, 1))
        loss_v = ((v - zt) ** 2).mean()
        loss = loss_pi + loss_v
        loss.backward()
        self.grads = {i: self.w[i].grad.detach().clone() for i in TRAINABLE}
        for i in TRAINABLE:
            self.w[i].requires_grad_(False)
        self.outputs = dict(p=p.detach().numpy().reshape(B, n * n), v=v.detach().numpy()[:, 0])
        return loss.item(), loss_pi.item(), loss_v.item()

    def apply(self, grads=None):
        """one tf.keras Adam step (+ the BN moving-statistics update of the forward pass just run)"""
        grads = grads if grads is not None else self.grads
        self.step += 1
        t, b1, b2 = self.step, 0.9, 0.999
        lr_t = self.lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        for i in TRAINABLE:
            g = grads[i]
            if self.clip and self.clip > 0:
                g = torch.clamp(g, -self.clip, self.clip)
            self.m[i] = b1 * self.m[i] + (1 - b1) * g
            self.v[i] = b2 * self.v[i] + (1 - b2) * g * g
            self.w[i] = self.w[i].detach() - lr_t * self.m[i] / (torch.sqrt(self.v[i]) + 1e-7)
        for i, s in self._new_stats.items():
            self.w[i] = s
        self._new_stats = {}

    def weights(self):
        return [a.detach().numpy().copy() for a in self.w]
