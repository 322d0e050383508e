import sys, numpy as np
sys.path.insert(0, "/root/repo")
from othellozero_amd.NNet import NNetWrapper
net = NNetWrapper((8, 8), num_channels_1=512, max_batch=4096, seed=0, precision="f16x2")
rs = np.random.RandomState(0)
valid = np.uint64(0xFFFFFFFFFFFFFFFF)
own = rs.randint(0, 2**63, size=4096, dtype=np.uint64); opp = rs.randint(0, 2**63, size=4096, dtype=np.uint64) & ~own
for cnt in (3328, 3414, 3500, 3600, 3736, 3840, 3968, 4096):
    for _ in range(3): net.predict_batch(own[:cnt], opp[:cnt])
    net.profile(2); net.profile_kernels(reset=True)
    for _ in range(20): net.predict_batch(own[:cnt], opp[:cnt])
    k = net.profile_kernels(); net.profile(0)
    c3 = k["conv3"][0] / k["conv3"][1]; c4 = k["conv4"][0] / k["conv4"][1]
    print(cnt, "conv3 %.3f ms (%.1f blocks-rounds) conv4 %.3f ms" % (c3, (cnt * 36 / 192) * 2 / 256, c4), "us/leaf conv3 %.4f" % (c3 * 1e3 / cnt))
