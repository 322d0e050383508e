"""Where to put the limit of the commit-time self-check: for healthy and badly conditioned networks, the error of both precisions against
the float64 oracle on test boards (E16, E32) next to what the self-check measures on the calibration positions (D = f16x2 vs f32)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import nn_numpy
    from othellozero_amd import _lib
    from othellozero_amd.NNet import NNetWrapper
    from othellozero_amd.weights import init_weights
    import test_gpu_parity as T
    cases = [None, "small_conv2_kernel_small_bn3_variance", "tiny_activations", "weights_span_2^20_by_output_channel", "weights_span_2^20_unstructured",
             "dense_layers_rescaled"]
    for n, C_, mb in ((8, 256, 64), (8, 512, 64), (6, 512, 4), (8, 512, 4096)):
        for seed in (11, 21):
            for case in cases:
                if case is not None and (seed != 21 or mb == 4096):
                    continue
                base = init_weights(n, seed=seed, channels=C_, randomize_all=True)
                for i in (36, 38):
                    base[i] = base[i] * 4.0
                w = base if case is None else T._rescaled(base, case, C_)
                own, opp = T._boards(n, 48, seed=5)
                pi64, v64 = nn_numpy.forward(w, own, opp, n)
                out = {}
                for prec in ("f32", "f16x2"):
                    net = NNetWrapper.__new__(NNetWrapper)
                    try:
                        net = NNetWrapper((n, n), num_channels_1=C_, max_batch=mb, weights=w, precision=prec) if prec == "f32" else None
                        if prec == "f16x2":
                            # construct without the enforcing self-check: measure only
                            import ctypes as C
                            net = NNetWrapper.__new__(NNetWrapper)
                            from othellozero_amd.NNet import _NetHandle, NeuralNets
                            _NetHandle.__init__(net)
                            net.board_size_x = net.board_size_y = n; net.num_channels = C_; net.max_batch = mb; net.in_channels = 2
                            net.network_type = NeuralNets.ONN; net.precision = prec
                            lib = _lib.load()
                            _lib.check(lib.oz_net_create(C.byref(net._h), n, C_, mb))
                            _lib.check(lib.oz_net_set_precision(net._h, 1))
                            _lib.check(lib.oz_net_set_option(net._h, _lib.NET_OPT_SELF_CHECK, 2))
                            net.set_weights(w)
                        pi, v = net.predict_batch(own, opp)
                        out[prec] = (np.abs(pi.reshape(48, -1) - pi64).max(), np.abs(v - v64).max())
                        if prec == "f16x2":
                            out["D"] = net.self_check()
                    except _lib.OzError as e:
                        out[prec] = str(e)[:60]
                print(f"n={n} C={C_} mb={mb} seed={seed} case={case}: E32={out.get('f32')} E16={out.get('f16x2')} D={out.get('D')}", flush=True)


if __name__ == "__main__":
    main()
