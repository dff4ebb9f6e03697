"""
Synthetic dual-polarisation captures + a minimal SER counter for the measurement harness.

The reference's generator / impairment / BER stack (qampy/signals.py, qampy/core/impairments.py, qampy/core/resample.py,
qampy/core/ber_functions.py) cannot travel to the GPU box and is out of scope, so the
bench and the tests synthesise their inputs here.  Conventions follow the reference so that the recipes of SURVEY.md §8d
carry over:
  * noise:  sigma = sqrt(P)*10^(-snr/20)*sqrt(os), split equally over I and Q   (core/impairments.py:205, 230-233)
  * PMD:    R(-theta) diag(e^{-j w tau/2}, e^{+j w tau/2}) R(theta) in the frequency domain (core/impairments.py:94-104)
  * phase noise: Wiener process, variance 2*pi*linewidth/fs per sample, applied per transmitted mode before the
    polarisation mixing                                                        (core/impairments.py:155-158)
This is host-side numpy; it is never inside a timed region.
"""
import numpy as np

from . import theory
from .signals import SignalQAM


def _rrc_frequency_response(n, os, beta):
    """Root-raised-cosine |H(f)| sampled on an n-point FFT grid, symbol rate = fs/os."""
    f = np.fft.fftfreq(n) * os          # frequency in units of the symbol rate
    af = np.abs(f)
    H = np.zeros(n)
    lo = (1 - beta) / 2
    hi = (1 + beta) / 2
    H[af <= lo] = 1.0
    if beta > 0:
        roll = (af > lo) & (af <= hi)
        H[roll] = np.sqrt(0.5 * (1 + np.cos(np.pi / beta * (af[roll] - lo))))
    return H


def make_capture(M, nsym, nmodes=2, os=2, snr_db=None, theta=None, dgd=None, linewidth=0., fb=20e9, beta=0.1,
                 seed=1000, dtype=np.complex64, shift=0, symbols=None):
    """
    Build one impaired capture.

    Returns a :class:`SignalQAM` of shape ``(nmodes, nsym*os)`` sampled at ``os*fb`` whose ``.symbols`` holds the
    transmitted unit-power Gray-mapped symbols ``(nmodes, nsym)``.

    ``shift`` circularly delays the waveform by that many samples (used by the data-aided tests, which need symbol
    ``i`` under the centre tap of window ``i``, cf. test/test_equalisation.py:109-110).
    """
    rng = np.random.default_rng(seed)
    alphabet = theory.coded_symbols_qam(M, dtype=np.complex128)
    idx = rng.integers(0, M, size=(nmodes, nsym))
    syms = alphabet[idx] if symbols is None else np.asarray(symbols, dtype=np.complex128)      # given symbols: cross-checks
    L = nsym * os
    fs = fb * os
    # zero-stuff to os samples/symbol and RRC-shape in the frequency domain (circular)
    up = np.zeros((nmodes, L), dtype=np.complex128)
    up[:, ::os] = syms
    H = _rrc_frequency_response(L, os, beta)
    x = np.fft.ifft(np.fft.fft(up, axis=1) * H, axis=1)
    x /= np.sqrt(np.mean(np.abs(x) ** 2, axis=1, keepdims=True))
    if linewidth:
        # transmitter-side Wiener phase noise, independent per mode like the reference's apply_phase_noise; it is applied
        # BEFORE the polarisation mixing so that a static 2x2 equaliser still separates the modes and the per-mode BPS
        # sees one phase process per output
        var = 2 * np.pi * linewidth / fs
        ph = np.cumsum(rng.normal(scale=np.sqrt(var), size=(nmodes, L)), axis=1)
        x = x * np.exp(1j * ph)
    if theta is not None and nmodes == 2:
        omega = 2 * np.pi * np.fft.fftfreq(L, d=1 / fs)
        c, s = np.cos(theta), np.sin(theta)
        X = np.fft.fft(x, axis=1)
        a = c * X[0] - s * X[1]
        b = s * X[0] + c * X[1]
        tau = 0. if dgd is None else dgd
        a = a * np.exp(-0.5j * omega * tau)
        b = b * np.exp(+0.5j * omega * tau)
        X0 = c * a + s * b
        X1 = -s * a + c * b
        x = np.fft.ifft(np.stack([X0, X1]), axis=1)
    if snr_db is not None:
        p = np.mean(np.abs(x) ** 2)
        sigma = np.sqrt(p) * 10 ** (-snr_db / 20) * np.sqrt(os)
        x = x + sigma * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape)) / np.sqrt(2)
    if shift:
        x = np.roll(x, shift, axis=1)
    return SignalQAM(np.ascontiguousarray(x.astype(dtype)), M, fb=fb, fs=fs, symbols=syms.astype(dtype),
                     coded_symbols=alphabet.astype(dtype))


def make_pilot_capture(M=256, frame_len=2 ** 16, seq_len=2 ** 10, ins_rat=32, nframes=3, nmodes=2, os=2, fb=24e9, snr_db=35, theta=np.pi / 5.6,
                       dgd=10e-12, linewidth=10e3, freq_off=40e6, modal_delay=700, frame_offset=12345, seed=9, dtype=np.complex128):
    """
    A capture for the pilot-based receiver (BASELINE config 5 shape): frames of ``frame_len`` symbols that start with a QPSK
    pilot sequence of ``seq_len`` symbols, then one QPSK phase pilot every ``ins_rat`` symbols among the M-QAM payload; the
    second mode is delayed by ``modal_delay`` symbols, the frame starts ``frame_offset`` symbols into the capture, a carrier
    frequency offset is applied after the usual impairments (:func:`make_capture`).  Returns a dict: ``E`` (nmodes, nframes *
    frame_len * os), ``pilots`` (nmodes, pilots per frame: sequence first), ``payload`` (nmodes, payload symbols per frame),
    ``alphabet``, geometry and rates - what :class:`qampy_amd.signals.PilotSignal` takes.
    """
    from .signals import PilotSignal
    rng = np.random.default_rng(seed)
    _, idx_dat, idx_pil = PilotSignal._cal_pilot_idx(frame_len, seq_len, ins_rat)
    pil_alpha, dat_alpha = theory.coded_symbols_qam(4, dtype=np.complex128), theory.coded_symbols_qam(M, dtype=np.complex128)
    pilots = pil_alpha[rng.integers(0, 4, (nmodes, int(idx_pil.sum())))]
    payload = dat_alpha[rng.integers(0, M, (nmodes, int(idx_dat.sum())))]
    frame = np.empty((nmodes, frame_len), np.complex128)
    frame[:, idx_pil] = pilots
    frame[:, idx_dat] = payload
    tx = np.tile(frame, nframes)
    for m in range(1, nmodes):
        tx[m] = np.roll(tx[m], m * modal_delay)
    cap = make_capture(M, tx.shape[1], nmodes=nmodes, os=os, snr_db=snr_db, theta=theta if nmodes == 2 else None, dgd=dgd, linewidth=linewidth,
                       fb=fb, beta=0.1, seed=seed, dtype=np.complex128, symbols=tx)
    E = np.roll(np.asarray(cap), os * frame_offset, axis=1)
    E = E * np.exp(2j * np.pi * freq_off / (fb * os) * np.arange(E.shape[1]))
    return dict(E=np.ascontiguousarray(E.astype(dtype)), pilots=pilots, payload=payload, alphabet=dat_alpha, M=M, fb=fb, fs=fb * os,
                frame_len=frame_len, seq_len=seq_len, ins_rat=ins_rat, nframes=nframes)


def make_capture_dev(M, nsym, nmodes=2, os=2, snr_db=None, theta=None, dgd=None, linewidth=0., fb=20e9, beta=0.1, seed=1000, E=None):
    """
    :func:`make_capture` on the GPU (``qh_synth_capture_c64_dev``, csrc/synth.hip): the capture never exists on the host.
    Same impairment conventions; time-domain filters instead of FFTs and a counter-based generator instead of numpy's, so
    the waveforms agree statistically (and, for given symbols, to the filter truncation error - see the tests), not bit for bit.

    Returns ``dict(E=(nmodes, nsym*os) complex64, symbols=(nmodes, nsym) complex64, idx_tx=(nmodes, nsym) int32, alphabet=(M,))``
    of :class:`qampy_amd._lib.DeviceArray` plus ``fb, fs, M``.  ``E``: optional preallocated ``(nmodes, nsym*os)`` DeviceArray
    (e.g. one channel of a :class:`qampy_amd.pipeline.ChannelBank`) to synthesise into.
    """
    from . import _lib
    from ._lib import DeviceArray
    alphabet = np.ascontiguousarray(theory.coded_symbols_qam(M, dtype=np.complex64))
    d_al = DeviceArray.from_host(alphabet)
    if E is None:
        E = DeviceArray((nmodes, nsym * os), np.complex64)
    assert tuple(E.shape) == (nmodes, nsym * os) and np.dtype(E.dtype) == np.complex64
    sy = DeviceArray((nmodes, nsym), np.complex64)
    idx = DeviceArray((nmodes, nsym), np.int32)
    fs = fb * os
    _lib.call("qh_synth_capture_c64_dev", E.ptr, sy.ptr, idx.ptr, d_al.ptr, int(M), int(nmodes), int(nsym), int(os), float(beta),
              float(snr_db if snr_db is not None else 0.), int(snr_db is not None), float(theta if theta is not None else 0.),
              float((dgd or 0.) * fs), int(theta is not None and nmodes == 2), float(2 * np.pi * linewidth / fs if linewidth else 0.),
              int(seed))
    return dict(E=E, symbols=sy, idx_tx=idx, alphabet=d_al, alphabet_host=alphabet, fb=fb, fs=fs, M=M)


# ------------------------------------------------------------------------------------------------- SER harness
def normalise_and_center(E):
    """Per-mode mean removal and unit-power scaling (behaviour of qampy/helpers.py:46-58)."""
    E = np.atleast_2d(E)
    E = E - E.mean(axis=-1, keepdims=True)
    return E / np.sqrt(np.mean(np.abs(E) ** 2, axis=-1, keepdims=True))


def decide(E, alphabet, chunk=1 << 16):
    """Index of the nearest alphabet point for every sample (host numpy, harness only)."""
    E = np.asarray(E).ravel()
    out = np.empty(E.size, dtype=np.int32)
    for a in range(0, E.size, chunk):
        out[a:a + chunk] = np.argmin(np.abs(E[a:a + chunk, None] - alphabet[None, :]) ** 2, axis=1)
    return out


def count_symbol_errors(rx, tx_symbols, alphabet, max_lag=256, trim=0):
    """
    Symbol errors of one equalised mode against the transmitted sequences.

    The blind equaliser leaves a 4-fold rotation, an unknown delay and possibly swapped polarisations, so every
    (tx mode, rotation) hypothesis is tried; the delay is the peak of a circular cross-correlation restricted to
    ``|lag| <= max_lag``.  Returns ``(nerr, ncompared, tx_mode, rot, lag)`` of the best hypothesis.
    """
    rx = np.asarray(rx).ravel()
    if trim:
        rx = rx[trim:-trim]
    tx_symbols = np.atleast_2d(tx_symbols)
    n = rx.size
    nfft = 1 << int(np.ceil(np.log2(n + tx_symbols.shape[1])))
    RX = np.fft.fft(rx, nfft)
    best = None
    for m in range(tx_symbols.shape[0]):
        tx = tx_symbols[m]
        xc = np.fft.ifft(RX * np.conj(np.fft.fft(tx, nfft)))      # xc[l] = sum rx[i+l] conj(tx[i])
        lags = np.r_[0:max_lag + 1, nfft - max_lag - trim - 1:nfft]
        k = lags[np.argmax(np.abs(xc[lags]))]
        lag = k if k <= max_lag else k - nfft                      # rx[i] ~ tx[i - lag]
        rot = np.exp(-1j * np.round(np.angle(xc[k]) / (np.pi / 2)) * np.pi / 2)
        i0 = max(0, lag)
        i1 = min(n, tx.size + lag)
        if i1 <= i0:
            continue
        d_rx = decide(rx[i0:i1] * rot, alphabet)
        d_tx = decide(tx[i0 - lag:i1 - lag], alphabet)
        nerr = int(np.count_nonzero(d_rx != d_tx))
        cand = (nerr, i1 - i0, m, int(np.round(np.angle(rot) / (np.pi / 2))) % 4, int(lag))
        if best is None or cand[0] / cand[1] < best[0] / best[1]:
            best = cand
    return best


def cal_ser(rx, tx_symbols, alphabet, trim=0, max_lag=256):
    """Per-mode symbol-error rate of an equalised (and phase-recovered) signal; see :func:`count_symbol_errors`."""
    rx = np.atleast_2d(rx)
    out = []
    for r in rx:
        nerr, ncmp = count_symbol_errors(r, tx_symbols, alphabet, max_lag=max_lag, trim=trim)[:2]
        out.append(nerr / ncmp)
    return np.array(out)
