"""Kernel experiment harness (GPU box): time the OthelloNN forward and the conv2 launch at batch 4096.
usage: python tools/exp_net.py [precision] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from othellozero_amd.NNet import NNetWrapper
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(os.environ.get("EXP_B", "4096"))
net = NNetWrapper((8, 8), num_channels_1=512, max_batch=B, seed=0, precision=prec)
net.time_forward(B, 2)
net.profile(True)
ms = net.time_forward(B, iters)
c2, n = net.profile_read()
layer = net.profiled_layer()                 # 2 = conv2 GEMM, 3 = conv3 GEMM (conv1 + conv2 as the table gather-sum)
flop = B * 2 * (64 if layer == 2 else 36) * 4608 * 512
print(f"{prec} B={B}: forward {ms:.3f} ms  ({B*566428672/ms/1e9:.1f} TF-eq of the reference network)   conv{layer} {c2/n:.3f} ms ({flop/(c2/n)/1e9:.1f} TF-eq, n={n})", flush=True)
