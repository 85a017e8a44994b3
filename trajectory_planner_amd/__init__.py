"""MI355X-native batched back-end for the ViGO B-spline optimizer hot path and the min-snap
corridor collision checker of hanyujin02/trajectory_planner.

The product is the C-ABI library built from csrc/ (include/vigo.h); `vigo.Vigo` is the thin
Python host side used by tests and bench.py.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
