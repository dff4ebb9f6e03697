"""
Minimal signal object for the hot path.

QAMpy's signal classes (qampy/signals.py, 1952 lines) are out of scope.  The hot-path wrappers only
touch ``sig.os, sig.M, sig.fb, sig.fs, sig.coded_symbols, sig.symbols`` and ``sig.recreate_from_np_array(arr, fs=...)``
(qampy/equalisation.py:246-259, qampy/phaserec.py:92, qampy/signals.py:179-181, :209-220, :872-878), so this module
provides one duck-typed ndarray subclass carrying exactly these.  Any object exposing the same attributes - including a
real ``qampy.signals.SignalQAMGrayCoded`` - works with ``qampy_amd.equalisation`` / ``qampy_amd.phaserec``.
:class:`PilotSignal` does the same for the pilot receiver (the attributes of ``SignalWithPilots`` the receiver reads).
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


# ------------------------------------------------------------------------------------------------ pilot frames
_PATTRS = ("_M", "_fb", "_fs", "_coded_symbols", "_symbols", "_pilots", "_frame_len", "_pilot_seq_len", "_pilot_ins_rat",
           "_idx_dat", "_idx_pil", "_Mpilots", "_shiftfctrs", "_synctaps", "_foe")


class PilotSignal(np.ndarray):
    """
    Pilot-frame capture ``(nmodes, nsamples)``: the attributes of ``qampy.signals.SignalWithPilots`` that the pilot receiver
    reads (qampy/signals.py:1430-1760) - frame geometry, pilot sequence / phase pilots, ``shiftfctrs`` / ``synctaps`` after
    :meth:`sync2frame` - on a plain ndarray subclass.  A frame is ``pilot_seq_len`` pilot symbols followed by payload with
    one phase pilot every ``pilot_ins_rat`` symbols (:1532-1545).  Generation of such signals is out of scope; build one
    from arrays (a received capture plus the known pilots).

    Parameters
    ----------
    data : (nmodes, N) complex samples
    M : payload QAM order;  fb, fs : symbol / sampling rate
    frame_len, pilot_seq_len, pilot_ins_rat : frame geometry in symbols
    pilots : (nmodes, >= n_pilots_per_frame) complex, pilot sequence first then the phase pilots
    symbols : (nmodes, n_payload_per_frame) transmitted payload (optional, SER only);  Mpilots : pilot QAM order
    """

    def __new__(cls, data, M, fb, fs, frame_len, pilot_seq_len, pilot_ins_rat, pilots, symbols=None, Mpilots=4,
                coded_symbols=None):
        obj = np.atleast_2d(np.asarray(data)).view(cls)
        if not np.iscomplexobj(obj):
            raise ValueError("PilotSignal needs a complex array")
        obj._M, obj._fb, obj._fs, obj._Mpilots = int(M), fb, fs, int(Mpilots)
        obj._frame_len, obj._pilot_seq_len, obj._pilot_ins_rat = int(frame_len), int(pilot_seq_len), pilot_ins_rat
        idx, obj._idx_dat, obj._idx_pil = cls._cal_pilot_idx(frame_len, pilot_seq_len, pilot_ins_rat)
        obj._pilots = np.atleast_2d(np.asarray(pilots))
        if obj._pilots.shape[1] < np.count_nonzero(obj._idx_pil):
            raise ValueError("a frame holds %d pilots, got %d" % (np.count_nonzero(obj._idx_pil), obj._pilots.shape[1]))
        obj._symbols = None if symbols is None else np.atleast_2d(np.asarray(symbols))
        obj._coded_symbols = theory.coded_symbols_qam(M, dtype=obj.dtype) if coded_symbols is None else np.asarray(coded_symbols)
        obj._shiftfctrs = obj._synctaps = None
        obj._foe = 0
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        for a in _PATTRS:
            setattr(self, a, getattr(obj, a, None))

    @staticmethod
    def _cal_pilot_idx(frame_len, pilot_seq_len, pilot_ins_rat):
        """Positions of pilots / payload inside a frame (behaviour of qampy/signals.py:1532-1545)."""
        where = np.arange(frame_len)
        behind = where - pilot_seq_len                   # position counted from the end of the pilot sequence
        idx_pil = behind < 0
        if pilot_ins_rat:
            if (frame_len - pilot_seq_len) % pilot_ins_rat:
                raise ValueError("Frame without pilot sequence divided by pilot rate needs to be an integer")
            idx_pil = idx_pil | (behind % pilot_ins_rat == 0)
        idx = where
        return idx, ~idx_pil, idx_pil

    M = property(lambda self: self._M)
    Mpilots = property(lambda self: self._Mpilots)
    fb = property(lambda self: self._fb)
    fs = property(lambda self: self._fs)
    os = property(lambda self: int(self.fs / self.fb))
    coded_symbols = property(lambda self: self._coded_symbols)
    symbols = property(lambda self: self._symbols)
    pilots = property(lambda self: self._pilots)
    pilot_seq = property(lambda self: self._pilots[:, :self._pilot_seq_len])
    ph_pilots = property(lambda self: self._pilots[:, self._pilot_seq_len:])
    frame_len = property(lambda self: self._frame_len)
    nframes = property(lambda self: self.shape[-1] // (self.os * self.frame_len))
    idx_payload = property(lambda self: np.tile(self._idx_dat, self.nframes)[:self.shape[-1]])
    idx_pilots = property(lambda self: np.tile(~self._idx_dat, self.nframes)[:self.shape[-1]])

    @property
    def shiftfctrs(self):
        return self._shiftfctrs

    @shiftfctrs.setter
    def shiftfctrs(self, value):
        self._shiftfctrs = value

    @property
    def synctaps(self):
        return self._synctaps

    @synctaps.setter
    def synctaps(self, value):
        self._synctaps = value

    def recreate_from_np_array(self, arr, **kwargs):
        out = np.atleast_2d(np.asarray(arr)).view(type(self))
        for a in _PATTRS:
            setattr(out, a, getattr(self, a))
        if "fb" in kwargs and "fs" not in kwargs:
            kwargs["fs"] = self.os * kwargs["fb"]
        for k, v in kwargs.items():
            setattr(out, "_" + k if "_" + k in _PATTRS else k, v)
        return out

    def sync2frame(self, returntaps=False, **kwargs):
        """Find the start of the pilot sequence per mode, reorder the modes accordingly and store ``shiftfctrs`` /
        ``synctaps`` (behaviour of qampy/signals.py:1709-1745; all search windows train in ONE launch)."""
        from .core import pilotbased_receiver
        search = dict(adaptive_stepsize=True, Niter=10, method="cma", Ntaps=17, mu=5e-3)
        search.update(kwargs)
        ntaps = search.pop("Ntaps")
        shift, foe, order, taps, found = pilotbased_receiver.frame_sync(np.asarray(self), np.asarray(self.pilot_seq), self.os, frame_len=self.frame_len,
                                                                        M_pilot=self.Mpilots, Ntaps=ntaps, **search)
        period = self.frame_len * self.os                 # shifts are positions inside one frame period
        self[:, :] = np.asarray(self)[order, :]
        self.shiftfctrs = np.where(np.asarray(shift) < 0, np.asarray(shift) + period, shift)[order]
        self.synctaps, self._foe = ntaps, foe
        return (taps, found) if returntaps else found

    def corr_foe(self, additional_foe=0):
        """Remove the coarse frequency offset found by :meth:`sync2frame` (qampy/signals.py:1747-1750)."""
        from .core import phaserecovery
        foe_off = np.ones(np.asarray(self._foe).shape) * (np.mean(self._foe) + additional_foe)
        self._foe = 0
        self[:, :] = phaserecovery.comp_freq_offset(np.asarray(self), foe_off, self.os)

    def _frame_mask(self, per_frame, frames):
        frames = np.arange(self.nframes) if frames is None else np.atleast_1d(frames)
        idx = np.zeros(self.shape[-1], dtype=bool)
        for i in frames:
            idx[i * self.frame_len:(i + 1) * self.frame_len] = per_frame[:max(0, min(self.frame_len, self.shape[-1] - i * self.frame_len))]
        return idx

    def get_data(self, frames=None):
        """Payload symbols of a frame-aligned, 1 sample/symbol signal (qampy/signals.py:1753-1781)."""
        return np.asarray(self)[:, self._frame_mask(self._idx_dat, frames)].copy()

    def extract_pilots(self, frames=None):
        """Pilot symbols of a frame-aligned, 1 sample/symbol signal (qampy/signals.py:1783-1804)."""
        return np.asarray(self)[:, self._frame_mask(self._idx_pil, frames)].copy()

    def cal_ser(self, frames=None):
        """Symbol error rate of the payload against ``symbols`` (aligned by construction once the frame is synced)."""
        from . import synth
        if self._symbols is None:
            raise ValueError("no transmitted payload attached")
        frames = np.arange(self.nframes) if frames is None else np.atleast_1d(frames)
        rx = self.get_data(frames)
        tx = np.tile(self._symbols, len(frames))
        return np.array([np.mean(synth.decide(r, self._coded_symbols) != synth.decide(t[:r.size], self._coded_symbols)) for r, t in zip(rx, tx)])
