"""MI355X-native (gfx950) implementation of the discrete edge-diffusion hot path of
chinmay5/semantically_consistent_graph_generation: GraphTransformer denoiser, absorbing-edge reverse step with
symmetric categorical sampling, structural extra features and the T-step sampling loop.

The Python modules keep the reference's class and method names and call hand-written HIP
kernels through the C ABI declared in include/scgg.h (libscgg.so, bound with ctypes in _lib.py).  There is no CPU
fallback: importing the compute modules without the built library raises.
"""
from .placeholder import PlaceHolder  # noqa: F401

__all__ = ["PlaceHolder"]
