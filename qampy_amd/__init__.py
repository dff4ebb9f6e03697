"""
qampy_amd - MI355X-native (gfx950) implementation of QAMpy's adaptive-equaliser + blind-phase-search hot path.

    qampy_amd.equalisation.{equalise_signal, dual_mode_equalisation, apply_filter}     <- qampy.equalisation
    qampy_amd.phaserec.bps                                                              <- qampy.phaserec
    qampy_amd.core.equalisation / qampy_amd.core.phaserecovery                          <- qampy.core.*
    qampy_amd.core.equalisation.hip_equalisation / qampy_amd.core.hip_dsp               <- the two pythran extensions

Python host code calling hand-written HIP kernels through the ctypes C ABI of include/qampy_hip.h.  No PyTorch, no
Triton, no CPU fallback.
"""
from . import core, equalisation, phaserec, signals, theory  # noqa: F401
from ._lib import set_default_tier, get_default_tier  # noqa: F401   (tier a = exact recurrence, the default; tier b = parallel in time)

__version__ = "0.1.0"
