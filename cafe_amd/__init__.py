"""cafe_amd -- MI355X-native likelihood engine for CAFE's per-family birth-death hot path.

The product is the HIP library behind include/cafehip.h (cafe_amd/csrc); this package is the
thin host-side mirror used by tests and bench.py.
"""
from ._lib import CafeHipError, load, lib_path  # noqa: F401
from .engine import Engine, FamilySizeRange, init_family_size  # noqa: F401

__all__ = ["Engine", "FamilySizeRange", "init_family_size", "CafeHipError", "load", "lib_path"]
