"""
Drop-in for the BPS entry points of the compiled module ``qampy.core.pythran_dsp`` on MI355X.

``bps`` and ``select_angles`` keep the signatures of the ``#pythran export`` lines qampy/core/pythran_dsp.py:45-46 and
:133-136; the kernels live in qampy_amd/csrc/bps.hip.
"""
import numpy as np

from .. import _lib


def bps(E, testangles, symbols, N):
    """
    Blind phase search: int32 index of the best test angle for every symbol of ONE mode (pythran_dsp.py:45-85 with
    select_angle_index :26-42).  ``testangles`` is ``(1, A)`` (one grid) or ``(L, A)`` (per-symbol grid).
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    if E.ndim != 1 or not np.iscomplexobj(E):
        raise TypeError("bps works on a 1-d complex array")
    E = np.ascontiguousarray(E)
    if testangles.ndim != 2 or testangles.dtype != rt:
        raise TypeError("testangles must be a 2-d %s array" % np.dtype(rt).name)
    symbols = np.ascontiguousarray(symbols)
    if symbols.dtype != ct:
        raise TypeError("symbols must be %s" % np.dtype(ct).name)
    testangles = np.ascontiguousarray(testangles)
    p, A = testangles.shape
    L = E.shape[0]
    if not (p == 1 or p == L):
        raise ValueError("p must be either 1 or the length of the input signal")
    idx = np.zeros(L, dtype=np.int32)
    _lib.call("qh_bps_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), L, _lib.ptr(testangles), p, A, _lib.ptr(symbols),
              symbols.size, int(N), _lib.ptr(idx))
    return idx


def select_angles(angles, idx):
    """``angles[0, idx[i]]`` (one grid) or ``angles[i, idx[i]]`` (per-symbol grid), pythran_dsp.py:133-153."""
    suf, rt, ct = _lib.suffix(angles.dtype)
    angles = np.ascontiguousarray(angles)
    if angles.ndim != 2:
        raise TypeError("angles must be 2-d")
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    p, A = angles.shape
    L = idx.shape[0] if p <= 1 else p
    if idx.shape[0] < L:
        raise ValueError("a per-symbol grid of %d rows needs as many indices (got %d)" % (p, idx.shape[0]))
    if idx.size and (idx.max() >= A or idx.min() < 0):
        raise ValueError("angle index out of range")
    out = np.zeros(L, dtype=rt)
    _lib.call("qh_select_angles_f" + suf, _lib.ptr(angles), p, A, _lib.ptr(idx), L, _lib.ptr(out))
    return out


def test_angle_grid(Mtestangles, rt):
    """The ``(1, A)`` grid of the host layer: ``np.linspace(-pi/4, pi/4, A, endpoint=False)`` formed in double and cast
    (qampy/core/phaserecovery.py:145)."""
    return np.linspace(-np.pi / 4, np.pi / 4, int(Mtestangles), endpoint=False, dtype=rt).reshape(1, -1)


def bps_recover_dev(E, Mtestangles, symbols, N, idx, ph, Eout, angles=None, part=0, nparts=1):
    """
    Device-resident carrier recovery of all modes at once: BPS index, grid look-up, ``np.unwrap`` of the interior and
    de-rotation (host layer qampy/core/phaserecovery.py:145-159) without leaving HBM.  All arguments except the integers
    are DeviceArrays: E, Eout (nmodes, L) complex; ph (nmodes, L) real; idx (nmodes, L) int32; ``angles`` the (A,) grid of
    :func:`test_angle_grid` (``None``: formed on the device in the signal's precision).

    ``part`` / ``nparts``: the work in ``nparts`` calls (``qh_bps_recover_part_*_dev``) - parts ``0 .. nparts - 2`` search a run of the signal each,
    the last one searches the rest, unwraps and de-rotates; same results, see ``ResidentReceiver.run(overlap=True)`` for what it is good for.
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    nm, L = E.shape
    if nparts == 1:
        _lib.call("qh_bps_recover_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, nm, L, angles.ptr if angles is not None else None,
                  int(Mtestangles), symbols.ptr, int(np.prod(symbols.shape)), int(N), idx.ptr, ph.ptr, Eout.ptr)
    else:
        _lib.call("qh_bps_recover_part_c" + ("64" if suf == "32" else "128") + "_dev", E.ptr, nm, L, angles.ptr if angles is not None else None,
                  int(Mtestangles), symbols.ptr, int(np.prod(symbols.shape)), int(N), idx.ptr, ph.ptr, Eout.ptr, int(part), int(nparts))


def bps_recover(E, Mtestangles, symbols, N):
    """
    Carrier recovery of every row of ``E (nmodes, L)`` in one go: upload once, blind phase search + unwrap + de-rotation in
    HBM, ``(Eout, ph)`` back.  What the host layer of the reference does mode by mode with three compiled / numpy passes.
    """
    suf, rt, ct = _lib.suffix(E.dtype)
    if E.ndim != 2 or not np.iscomplexobj(E):
        raise TypeError("bps_recover works on a 2-d complex array")
    symbols = np.ascontiguousarray(symbols)
    if symbols.dtype != ct:
        raise TypeError("symbols must be %s" % np.dtype(ct).name)
    nm, L = E.shape
    if L == 0:
        return np.zeros((nm, 0), ct), np.zeros((nm, 0), rt)
    D = _lib.DeviceArray
    E = np.ascontiguousarray(E)
    dsy, dang = D.from_host(symbols), D.from_host(test_angle_grid(Mtestangles, rt))
    dE, idx, ph, out = D((nm, L), ct), D((nm, L), np.int32), D((nm, L), rt), D((nm, L), ct)
    if E.nbytes < _lib.PINNED_MIN_BYTES:
        dE.set(E)
        bps_recover_dev(dE, Mtestangles, dsy, N, idx, ph, out, angles=dang)
        return out.to_host(), ph.to_host()
    # Rows (modes) are independent: row r + 1 goes up (stream 1) while row r is searched (stream 0) and row r - 1 comes back (stream 2, into
    # pooled pinned memory) - PCIe carries both directions at once; all searches on ONE stream (they share the library's scratch buffers).
    h_out, h_ph = _lib.pinned_empty((nm, L), ct), _lib.pinned_empty((nm, L), rt)
    ev_up, ev_done = [_lib.Event() for _ in range(nm)], [_lib.Event() for _ in range(nm)]
    try:
        for r in range(nm):
            _lib.call("qh_use_stream", 1)
            _lib.call("qh_memcpy_h2d_async", dE.row(r).ptr, E[r].ctypes.data, E[r].nbytes)
            ev_up[r].record()
            _lib.call("qh_use_stream", 0)
            _lib.call("qh_stream_wait_event", ev_up[r].ptr)
            bps_recover_dev(_row2d(dE, r), Mtestangles, dsy, N, _row2d(idx, r), _row2d(ph, r), _row2d(out, r), angles=dang)
            ev_done[r].record()
            _lib.call("qh_use_stream", 2)
            _lib.call("qh_stream_wait_event", ev_done[r].ptr)
            _lib.call("qh_memcpy_d2h_async", h_out[r].ctypes.data, out.row(r).ptr, h_out[r].nbytes)
            _lib.call("qh_memcpy_d2h_async", h_ph[r].ctypes.data, ph.row(r).ptr, h_ph[r].nbytes)
    finally:
        _lib.call("qh_use_stream", 0)
    _lib.sync()
    return h_out, h_ph


def _row2d(a, r):
    """Row ``r`` of a 2-d DeviceArray as a (1, L) view."""
    v = a.row(r)
    v.shape = (1,) + tuple(v.shape)
    return v


def comp_freq_offset(E, freq_offset, os=1):
    """``E[k, n] * exp(-2j pi (n + 1) freq_offset[k] / os)`` for every row (qampy/core/phaserecovery.py:435-473) on the device."""
    suf, rt, ct = _lib.suffix(E.dtype)
    E = np.ascontiguousarray(E)
    if E.ndim != 2 or not np.iscomplexobj(E):
        raise TypeError("comp_freq_offset works on a 2-d complex array")
    fo = np.ascontiguousarray(np.broadcast_to(np.asarray(freq_offset, dtype=np.float64).reshape(-1), (E.shape[0],)) if np.size(freq_offset) == 1
                              else np.asarray(freq_offset, dtype=np.float64).reshape(-1))
    if fo.size != E.shape[0]:
        raise ValueError("one frequency offset per mode (or one for all)")
    out = np.empty_like(E)
    _lib.call("qh_comp_freq_offset_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), E.shape[0], E.shape[1], _lib.ptr(fo), int(os), _lib.ptr(out))
    return out


def pilot_phase_trace(E, knots, knot_phase):
    """Linear interpolation of the pilot phases ``knot_phase (nmodes, nk)`` at the symbol positions ``knots`` to every symbol of ``E
    (nmodes, L)`` (np.interp) and its removal, on the device: ``(E * exp(-1j trace), trace)``, the trace in E's complex dtype as the
    reference returns it (qampy/core/pilotbased_receiver.py:318-327)."""
    suf, rt, ct = _lib.suffix(E.dtype)
    E = np.ascontiguousarray(E)
    if E.ndim != 2 or not np.iscomplexobj(E):
        raise TypeError("pilot_phase_trace works on a 2-d complex array")
    knots = np.ascontiguousarray(knots, dtype=np.int64)
    kph = np.ascontiguousarray(knot_phase, dtype=np.float64)
    if kph.shape != (E.shape[0], knots.size) or knots.size < 1:
        raise ValueError("one phase per mode and knot")
    big = E.nbytes >= _lib.PINNED_MIN_BYTES                       # results through the pinned pool: one DMA at the link rate each
    out = _lib.pinned_empty(E.shape, E.dtype) if big else np.empty_like(E)
    trace = _lib.pinned_empty(E.shape, E.dtype) if big else np.empty_like(E)
    _lib.call("qh_pilot_phase_trace_c" + ("64" if suf == "32" else "128"), _lib.ptr(E), E.shape[0], E.shape[1], _lib.ptr(knots), _lib.ptr(kph), knots.size,
              _lib.ptr(out), _lib.ptr(trace))
    return out, trace
