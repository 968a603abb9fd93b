"""MI355X-native message-passing engine for the GraphNet classifier's forward hot path.

The package mirrors the reference's module surface for that path only
(``models/GNN.py`` -> :mod:`graphnet_classifier_amd.GNN`, ``models/MLP.py`` ->
:mod:`graphnet_classifier_amd.MLP`); all arithmetic runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in ``include/gnc_hip.h``.
"""
__version__ = "0.1.0"
