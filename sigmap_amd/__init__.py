"""sigmap_amd -- MI355X-native streaming signal-mapping hot path (Sigmap drop-in).

Python here is plumbing only: ctypes bindings over the two C-ABI libraries
(include/sigmap_hip.h: HIP kernels; include/sigmap_host.h: host-side index build / PAF
logic), a synthetic-data generator, and the batch driver used by bench.py and the tests.
There is no CPU implementation of the hot path in this package: if libsigmap_hip.so is
missing or no GPU is present, the mapping entry points raise.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi", "synth", "pipeline"]
