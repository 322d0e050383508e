"""Cycle budget of the ping-pong conv loop (diagnostic build: tools/build_variant.sh stamps -DH2PP_STAMPS).
Prints, per wave of block 0 of the LAST conv launch (conv4 at 8x8... use --layer conv2 to stop after conv2 via a 1-layer
timing call), the cycles spent in: L work | DMA wait | mid barrier | LDS wait | MFMA issue | closing barrier."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OZ_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "othellozero_amd", "lib_stamps", "libothellozero_amd.so"))
import numpy as np
from othellozero_amd import _lib
from othellozero_amd.NNet import NNetWrapper
B = 4096
net = NNetWrapper((8, 8), num_channels_1=512, max_batch=B, seed=0, precision="f16x2")
net.time_forward(B, 3)
out = (C.c_ulonglong * 64)()
lib = _lib.load()
lib.oz_debug_h2_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.oz_debug_h2_stamps(out) == 0
a = np.array(out[:48], dtype=np.float64).reshape(8, 6)
clk = np.array(out[48:], dtype=np.float64).reshape(8, 2)
print('in-kernel clock (s_memtime / s_memrealtime x 100 MHz):', ' '.join(f'{c[0] / c[1] * 0.1:.3f}' for c in clk), 'GHz; loop', clk[0, 1] / 100, 'us')
names = ["L work", "DMA wait", "mid barrier", "LDS wait", "MFMA issue", "close barrier"]
print("block 0 of the last ping-pong launch (conv4: 144 k-tiles x 4 phases); cycles per phase, per wave")
tot = a.sum(axis=1)
for w in range(8):
    print(f"wave {w} (row {w // 4}):", "  ".join(f"{n} {a[w, i] / 576:7.1f}" for i, n in enumerate(names)), f"  total/phase {tot[w] / 576:7.1f}")
