"""MI355X-native normal-equation engine for JAICOV-style bundle adjustment (host-side Python plumbing).

The compute path is the HIP library ``csrc/libjaicov_neq.so`` behind the C ABI of ``include/jaicov_neq.h``; this
package only flattens problems, drives the engine through ctypes and wires multi-GPU runs through torch.distributed.
There is no CPU fallback: if the HIP library is missing or no gfx950 device is present the engine raises.
"""
from . import numbering, problem, scene  # noqa: F401
from .problem import FlatProblem  # noqa: F401

__all__ = ["FlatProblem", "numbering", "problem", "scene"]
