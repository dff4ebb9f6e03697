"""
Minimal signal object for the hot path.

QAMpy's signal classes (qampy/signals.py, 1952 lines) are out of scope.  The hot-path wrappers only
touch ``sig.os, sig.M, sig.fb, sig.fs, sig.coded_symbols, sig.symbols`` and ``sig.recreate_from_np_array(arr, fs=...)``
(qampy/equalisation.py:246-259, qampy/phaserec.py:92, qampy/signals.py:179-181, :209-220, :872-878), so this module
provides one duck-typed ndarray subclass carrying exactly these.  Any object exposing the same attributes - including a
real ``qampy.signals.SignalQAMGrayCoded`` - works with ``qampy_amd.equalisation`` / ``qampy_amd.phaserec``.
"""
import numpy as np

from . import theory

_ATTRS = ("_M", "_fb", "_fs", "_coded_symbols", "_symbols")


class SignalQAM(np.ndarray):
    """
    2-D complex array ``(nmodes, nsamples)`` with QAM metadata.

    Parameters
    ----------
    data : array_like (nmodes, N) complex
    M : QAM order
    fb : symbol rate, fs : sampling rate (``os = int(fs/fb)`` as in signals.py:179-181)
    symbols : transmitted symbol sequence (nmodes, nsym) the capture is based on (used by data-aided methods / SER)
    coded_symbols : alphabet in Gray-label order; generated from ``M`` when omitted
    """

    def __new__(cls, data, M, fb=1., fs=None, symbols=None, coded_symbols=None):
        obj = np.atleast_2d(np.asarray(data)).view(cls)
        if not np.iscomplexobj(obj):
            raise ValueError("SignalQAM needs a complex array")
        obj._M = int(M)
        obj._fb = fb
        obj._fs = fb if fs is None else fs
        if coded_symbols is None:
            coded_symbols = theory.coded_symbols_qam(M, dtype=obj.dtype)
        obj._coded_symbols = np.asarray(coded_symbols)
        obj._symbols = None if symbols is None else np.atleast_2d(np.asarray(symbols))
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        for a in _ATTRS:
            setattr(self, a, getattr(obj, a, None))

    # ---- the attributes the wrappers read
    @property
    def M(self):
        return self._M

    @property
    def fb(self):
        return self._fb

    @property
    def fs(self):
        return self._fs

    @property
    def os(self):
        return int(self.fs / self.fb)

    @property
    def coded_symbols(self):
        return self._coded_symbols

    @property
    def symbols(self):
        return self._symbols

    def recreate_from_np_array(self, arr, **kwargs):
        """Re-wrap a plain array with this signal's metadata (behaviour of signals.py:209-220)."""
        out = np.asarray(arr).view(type(self))
        for a in _ATTRS:
            setattr(out, a, getattr(self, a))
        if "fb" in kwargs and "fs" not in kwargs:
            kwargs["fs"] = self.os * kwargs["fb"]
        for k, v in kwargs.items():
            if "_" + k in _ATTRS:
                k = "_" + k
            setattr(out, k, v)
        return out
