"""noble-curves hot path on MI355X (gfx950): batch EC scalar multiplication, ed25519 batch
verification and Pippenger MSM behind the reference's Point / pippenger API names.

Host-side mirror of the reference interface for the hot path; all arithmetic
runs in hand-written HIP kernels inside libncg.so (C ABI: include/ncg.h).  There is no CPU
fallback: importing the engine without the built library, or using it without a GPU, raises.
"""
from ._native import Engine, NativeError, get_engine, lib_path  # noqa: F401

__all__ = ["Engine", "NativeError", "get_engine", "lib_path"]
__version__ = "0.1.0"
