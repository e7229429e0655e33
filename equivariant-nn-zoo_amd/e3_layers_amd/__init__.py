"""e3_layers_amd — MI355X-native tensor-product message passing behind the e3_layers module API.

Sub-packages mirror the reference's ``e3_layers`` layout for the hot path only:
``o3`` (irreps algebra, Wigner 3j), ``backend`` (ctypes binding of csrc/libe3k.so + autograd
glue), ``nn`` (layer modules), ``data`` (Data/Batch, edge construction), ``configs``
(model-config trees of the BASELINE configs), ``utils`` (factory helpers), ``run`` (graph-
parallel harness, VP-SDE helpers).
"""
__version__ = "0.1.0"
