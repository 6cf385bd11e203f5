"""locus_amd -- MI355X-native implementation of the CAVI-sweep hot path of `locus`.

Layout:
  csrc/            HIP kernels + the C ABI (include/locus_hip.h) -> liblocus_hip.so
  _lib.py          ctypes bindings of the C ABI (no fallback: fails without the .so / GPU)
  locus.py         locus(Y, X, p0_av, ...): the user-level entry (R/locus.R)
  core.py          coreLoop, coreBatch, ... : the reference's eight native routines
  device.py        device-resident state and kernels for the drivers
  update_vb.py, elbo.py, locus_core.py, locus_z_core.py, locus_struct_core.py,
  locus_bin_core.py   host side of the drivers (mirrors R/)
"""
from . import _lib  # noqa: F401
from .core import (coreBatch, coreLogitLoop, coreLoop, coreProbitBatch,  # noqa: F401
                   coreProbitLoop, coreStructLoop, coreZBatch, coreZLoop)

from .locus import locus  # noqa: F401,E402
from .cross_validate import set_cv  # noqa: F401,E402
from .prepare import set_blocks  # noqa: F401,E402
from .struct import set_struct  # noqa: F401,E402

__all__ = ["locus", "set_cv", "set_blocks", "set_struct", "coreLoop", "coreZLoop", "coreLogitLoop", "coreProbitLoop", "coreStructLoop",
           "coreBatch", "coreZBatch", "coreProbitBatch"]
