"""compairr_amd -- MI355X (gfx950) drop-in for CompAIRR's ``--matrix`` hot path.

The product is the C-ABI library ``compairr_amd/lib/libcompairr_hip.so``
(``include/compairr_hip.h``) and the C++11 host program ``bin/compairr``.  This
Python package only binds that C ABI (ctypes) for the test-suite and
``bench.py`` and holds the seeded synthetic-repertoire generator; it contains
no compute path of its own and no CPU fallback.
"""

from .sets import RepertoireSet  # noqa: F401
from .hip import HipOverlap, HipError, Options, Stats, library_path  # noqa: F401

__all__ = ["RepertoireSet", "HipOverlap", "HipError", "Options", "Stats", "library_path"]
