"""
Basic carrier-recovery API, mirror of ``qampy.phaserec.bps`` (qampy/phaserec.py:62-92) and ``bps_twostage`` (:24-60): the
alphabet is taken from the signal object and the signal subclass is preserved through the de-rotation.
"""
from . import core


def bps(E, Mtestangles, N, **kwargs):
    """Blind phase search on a signal object: ``(Eout, ph)``; see :func:`qampy_amd.core.phaserecovery.bps`."""
    return core.phaserecovery.bps(E, Mtestangles, E.coded_symbols, N, **kwargs)


def bps_twostage(E, Mtestangles, N, B=4, **kwargs):
    """Two-stage blind phase search on a signal object; see :func:`qampy_amd.core.phaserecovery.bps_twostage`."""
    return core.phaserecovery.bps_twostage(E, Mtestangles, E.coded_symbols, N, B=B, **kwargs)
