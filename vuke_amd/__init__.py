"""vuke_amd -- MI355X-native engine for vuke's seed -> key -> address -> match hot path.

The compute lives in vuke_amd/csrc (hand-written HIP for gfx950) behind the C ABI of
include/vuke_hip.h; this package is the thin host side.  Nothing here computes on the CPU.
"""
from . import _ffi
from ._ffi import Engine, VkError, spec  # noqa: F401

__all__ = ["Engine", "VkError", "spec", "_ffi"]
