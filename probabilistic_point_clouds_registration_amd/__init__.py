"""MI355X-native probabilistic point-cloud registration hot path.

    radius-NN correspondence search -> t/Gaussian soft-assignment weights -> weighted rigid solve

behind a C ABI (include/ppcr.h) implemented with hand-written HIP kernels for gfx950.
``_lib`` is the ctypes binding, ``registration`` mirrors the reference's
``prob_point_cloud_registration`` classes on top of it, ``synth`` generates the pinned
benchmark inputs, ``batch`` shards independent pairs over the GPUs of a node.

The HIP library is loaded on first use and there is no CPU fallback: without the built
extension (``python -m probabilistic_point_clouds_registration_amd.build``) or without a GPU every
compute entry point raises.
"""
__version__ = "0.1.0"
