"""Drop-in for the external package the reference imports at models/deformable_transformer.py:24
(`from deformable_attention import MSDeformAttn`; upstream: fundamentalvision/Deformable-DETR
models/ops, a CUDA extension).  Same class name, constructor, forward signature, parameter names
and `_reset_parameters()`; the arithmetic runs in libpoet_hip.so (gfx950)."""
from poet_amd.modules import MSDeformAttn  # noqa: F401

__all__ = ["MSDeformAttn"]
