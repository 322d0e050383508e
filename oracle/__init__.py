"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``othellozero_amd`` (the product) never does.

``oracle.lib()`` loads ``oracle/_build/liboz_oracle.so`` (``make -C oracle``),
the plain-C restatement of the reference's rules / search / drivers
(``oz_oracle.c``) and of the OthelloNN inference graph (``oz_oracle_nn.c``).
``oracle.nn_numpy`` is the float64 NumPy restatement of the same graph.

Parity status: rules / search / drivers are PINNED by ``tests/golden`` fixtures
generated from the reference's own Python (``tests/golden/gen_golden.py``).
The NN leg is UNPINNED by reference vectors (TensorFlow/Keras absent; see
DESIGN.md) and is cross-checked C-vs-NumPy-vs-torch instead.
"""
from .oracle import *  # noqa: F401,F403
